"""Golden vectors for path 1 host logic, produced by RUNNING the reference's own code (build container only).

* vidtome.npz  -- utils/VidToMe/vidtome/patch.py::compute_merge on CPU f32 over a 5-chunk chain (seed bank, local-src,
  bank-src, single frame), with the generator draws recorded; plus a chain of 8- / 6- / 16-frame chunks (several randframe rounds:
  8 -> 2 -> 1) and a chain with align_batch=False (per-sample matching).
* pipeline.npz -- Generator.temporal_denoise / ddim_sample / pred_noise (generate.py) and VidToMeGenerator.get_chunks
  (generate_utils.py) executed IN PLACE: the method source is pulled out of the reference files with `ast` at run time
  and exec'd against a stub `self` (generate.py cannot be imported here: diffusers / torch_scatter / cv2 are absent).
  Nothing is copied into the repo; only outputs are stored.
"""
import ast
import math
import os
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _method(path, cls, name, ns):
    src = open(path).read()
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for f in node.body:
                if isinstance(f, ast.FunctionDef) and f.name == name:
                    f.decorator_list = []
                    code = compile(ast.Module(body=[f], type_ignores=[]), path, "exec")
                    exec(code, ns)
                    return ns[name]
    raise KeyError(name)


def golden_vidtome(R):
    vt = R["vidtome"]
    patch = vt.patch
    out = {}
    N, C = 192, 32
    args = dict(max_downsample=2, generator=None, seed=1, batch_size=2, align_batch=True, merge_global=True,
                global_merge_ratio=0.5, local_merge_ratio=0.6, global_rand=0.5, target_stride=4)
    info = {"size": (12, 16), "args": args}
    mod = types.SimpleNamespace()
    mod.generator = torch.Generator(device="cpu").manual_seed(1)
    rng = np.random.default_rng(42)
    for ci, F in enumerate([4, 4, 3, 1, 2]):
        x = torch.from_numpy(rng.standard_normal((2 * F, N, C)).astype(np.float32))
        g2 = torch.Generator(device="cpu")
        g2.set_state(mod.generator.get_state())
        randf = int(torch.randint(0, min(4, F), torch.Size([1]), generator=g2)) if F > 1 else -1
        has_bank = getattr(mod, "global_tokens", None) is not None
        coin = float(torch.rand(1, generator=g2)) if has_bank else -1.0
        m, u, merged = patch.compute_merge(mod, x, info)
        T = merged.shape[1]
        ids = torch.arange(T, dtype=torch.float32)[None, :, None].repeat(2, 1, 1)
        unm = u(ids)[:F].reshape(-1).long()            # batch 0 frames -> joined positions
        out[f"c{ci}_F"], out[f"c{ci}_randf"], out[f"c{ci}_coin"], out[f"c{ci}_T"] = F, randf, coin, T
        out[f"c{ci}_merged"] = merged[:, ::7, ::5].numpy().copy()
        out[f"c{ci}_unm"] = unm.numpy()
        out[f"c{ci}_bank"] = mod.global_tokens[:, ::7, ::5].numpy().copy()
        out[f"c{ci}_bank_T"] = mod.global_tokens.shape[1]
        if not has_bank:
            assert torch.equal(m(x), merged)   # (with a bank m() itself raises in the reference; only merged_tokens is used)
    # ---- branches TC-Light's own configs leave idle: (m) chunks longer than target_stride are merged in several
    # randframe rounds, 8 -> 2 -> 1 frames, the unmerged tokens of earlier rounds joining the dst set (patch.py:43-56); (p) per-sample
    # matching, align_batch=False (merge.py:109-118, :422+).  Separate generators / RNG streams: the c* arrays above stay what they were.
    def chain(tag, frames, align_batch, seed):
        a2 = dict(args, generator=None, align_batch=align_batch)
        info2 = {"size": (12, 16), "args": a2}
        m2 = types.SimpleNamespace()
        m2.generator = torch.Generator(device="cpu").manual_seed(seed)
        r2 = np.random.default_rng(1000 + seed)
        for ci, F in enumerate(frames):
            x = torch.from_numpy(r2.standard_normal((2 * F, N, C)).astype(np.float32))
            g2 = torch.Generator(device="cpu")
            g2.set_state(m2.generator.get_state())
            randfs, cur = [], F
            while cur > 1:                               # the draws compute_merge is about to make, in its order
                ts = min(4, cur)
                rf = int(torch.randint(0, ts, torch.Size([1]), generator=g2))
                randfs.append(rf)
                cur = sum(1 for f in range(cur) if f % ts == rf)
            has_bank = getattr(m2, "global_tokens", None) is not None
            coin = float(torch.rand(1, generator=g2)) if has_bank else -1.0
            m, u, merged = patch.compute_merge(m2, x, info2)
            T = merged.shape[1]
            ids = torch.arange(T, dtype=torch.float32)[None, :, None].repeat(2, 1, 1)
            out[f"{tag}{ci}_F"], out[f"{tag}{ci}_randf"], out[f"{tag}{ci}_coin"], out[f"{tag}{ci}_T"] = F, np.array(randfs), coin, T
            out[f"{tag}{ci}_merged"] = merged[:, ::7, ::5].numpy().copy()
            out[f"{tag}{ci}_unm"] = u(ids).reshape(2, F * N).long().numpy()          # per sample (identical rows when aligned)
            out[f"{tag}{ci}_bank"] = m2.global_tokens[:, ::7, ::5].numpy().copy()
            out[f"{tag}{ci}_bank_T"] = m2.global_tokens.shape[1]
    chain("m", [8, 8, 6, 4, 16], True, 5)
    chain("p", [4, 4, 3, 1, 8], False, 6)
    np.savez_compressed(os.path.join(HERE, "vidtome.npz"), **out)
    print("vidtome.npz:", {k: v for k, v in out.items() if np.ndim(v) == 0})


def golden_pipeline(R):
    gu = R["general_utils"]
    from einops import rearrange
    out = {}
    # ---- get_chunks (generate_utils.py:174-205)
    ns = {"np": np, "torch": torch}
    get_chunks = _method(REF + "/utils/VidToMe/generate_utils.py", "VidToMeGenerator", "get_chunks", ns)
    for tag, flen in {"n8": 8, "n30": 30, "n300": 300, "w120": 120, "n3": 3}.items():
        fake = types.SimpleNamespace(chunk_size=4, merge_global=True, chunk_ord="mix", perm_div=4.0)
        np.random.seed(100 + flen)
        torch.manual_seed(100 + flen)
        chunks = get_chunks(fake, flen)
        np.random.seed(100 + flen)
        torch.manual_seed(100 + flen)
        rf = np.random.randint(0, 4)
        fl = np.random.rand()
        perm = torch.randperm(len(chunks))
        out[f"chunks_{tag}_draws"] = np.array([rf, fl])
        out[f"chunks_{tag}_perm"] = perm.numpy()
        out[f"chunks_{tag}_flat"] = torch.cat(chunks).numpy()
        out[f"chunks_{tag}_lens"] = np.array([len(c) for c in chunks])
    # ---- temporal_denoise (generate.py:241-284) with a marker pred_noise
    ns = {"np": np, "torch": torch, "math": math, "rearrange": rearrange,
          "adaptive_instance_normalization": gu.adaptive_instance_normalization}
    tden = _method(REF + "/generate.py", "Generator", "temporal_denoise", ns)
    for N in (8, 30, 64, 65, 127, 300):
        h, w = 3, 6
        g = np.random.default_rng(N)
        x = torch.from_numpy(g.standard_normal((N, 4, h, w)).astype(np.float32))
        noises = torch.from_numpy(g.standard_normal((N, 4, h, w)).astype(np.float32))
        calls = []

        def pred_noise(xt, cond, t, cc, batch_idx=None, sl_i=None):
            calls.append((int(sl_i), [int(c) for c in batch_idx], int(xt.shape[2])))
            return xt * 0.5 + cc * 0.25 + (sl_i + 1) * 0.01
        fake = types.SimpleNamespace(win_size_t=64, pred_noise=pred_noise,
                                     get_chunks=lambda n: [torch.arange(0, 2), torch.arange(2, n)])
        nt, nf = tden(fake, x, None, None, x * 2 + 1, 0.01 * 0.3, torch.zeros_like(x), noises.clone())
        out[f"tden_{N}_nt"], out[f"tden_{N}_nf"] = nt.numpy(), nf.numpy()
        out[f"tden_{N}_windows"] = np.array(sorted({(c[0], c[2]) for c in calls}))
    # ---- ddim_sample alpha schedule (generate.py:208-239)
    ds = _method(REF + "/generate.py", "Generator", "ddim_sample", {"torch": torch, "tqdm": lambda x, **k: x})
    alphas = []
    fake = types.SimpleNamespace(device="cpu", alpha_t=0.01, final_factor_t=0.01,
                                 scheduler=types.SimpleNamespace(timesteps=torch.arange(20).flip(0),
                                                                 step=lambda n, t, x, generator=None, return_dict=False: (x - 0.1 * n,)),
                                 pre_iter=lambda x, t: None, post_iter=lambda x, t: None, rng=None,
                                 get_chunks=lambda n: [torch.arange(n)],
                                 pred_noise=lambda x, c, t, cc, batch_idx=None: x * 0.5,
                                 temporal_denoise=lambda x, ct, t, cc, a, nt, n: (alphas.append(a) or nt, n))
    xo = ds(fake, torch.ones(3, 4, 2, 2), None, None, torch.zeros(3, 4, 2, 2))
    out["ddim_alphas"], out["ddim_x"] = np.array(alphas), xo.numpy()
    # ---- pred_noise (generate.py:288-352): CFG combine and batch layout with a marker unet
    pn = _method(REF + "/generate.py", "Generator", "pred_noise", {"torch": torch, "rearrange": rearrange})
    seen = {}

    def unet(inp, t, encoder_hidden_states=None, **kw):
        seen["in"], seen["text"], seen["cc"] = inp.clone(), encoder_hidden_states.clone(), kw["cross_attention_kwargs"]["concat_conds"]
        b = inp.shape[0] // 2
        return types.SimpleNamespace(sample=torch.cat([inp[:b] * 0.3, inp[b:] * 0.7 + 1.0]))
    fake = types.SimpleNamespace(use_pnp=False, use_depth=False, use_controlnet=False, model_key="iclight", unet=unet, guidance_scale=2.0)
    g = np.random.default_rng(9)
    x = torch.from_numpy(g.standard_normal((3, 4, 5, 6)).astype(np.float32))
    cond = torch.from_numpy(g.standard_normal((2, 7, 8)).astype(np.float32))
    res = pn(fake, x, cond, torch.tensor(5), x + 1)
    out["pn_out"], out["pn_text_rows"] = res.numpy(), seen["text"][:, 0, 0].numpy()
    np.savez_compressed(os.path.join(HERE, "pipeline.npz"), **out)
    print("pipeline.npz:", sorted(out)[:6], "...", len(out), "arrays")


GROUPS = {"vidtome": golden_vidtome, "pipeline": golden_pipeline}

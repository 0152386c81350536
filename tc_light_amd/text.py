"""Prompt embeddings (reference: generate.py:98-135 encode_prompt_inner / encode_prompt_pair).

Host plumbing, outside the two hot paths.  With a local CLIP ViT-L/14 text encoder directory (tokenizer + weights) the
reference's chunked scheme is reproduced with `transformers`: untruncated tokens -> 75-token chunks wrapped in BOS/EOS, padded
with EOS to 77 -> last_hidden_state per chunk -> shorter side tiled -> chunks concatenated along the sequence ->
cat([uncond, cond]).  Without weights (the build/bench images) a deterministic text-seeded stand-in of the right shape is returned.
"""
import hashlib
import math
import os
import warnings

import numpy as np
import torch

_ENC = {}


def _n_chunks_proxy(txt):
    return max(1, math.ceil(len(txt.replace(",", " , ").replace(".", " . ").split()) * 1.3 / 75))


def _inner(txt, dev, enc_dir):
    if enc_dir and os.path.isdir(enc_dir):
        if enc_dir not in _ENC:
            from transformers import CLIPTextModel, CLIPTokenizer
            _ENC[enc_dir] = (CLIPTokenizer.from_pretrained(enc_dir), CLIPTextModel.from_pretrained(enc_dir).to(dev).half().eval())
        tok, model = _ENC[enc_dir]
        ids = tok(txt, truncation=False, add_special_tokens=False)["input_ids"]
        L, bos, eos = tok.model_max_length, tok.bos_token_id, tok.eos_token_id
        chunks = [[bos] + ids[i:i + L - 2] + [eos] for i in range(0, max(len(ids), 1), L - 2)]
        chunks = [c[:L] + [eos] * (L - len(c)) for c in chunks]
        with torch.no_grad():
            return model(torch.tensor(chunks, device=dev)).last_hidden_state
    warnings.warn("CLIP text encoder not found -> deterministic text-seeded stand-in embeddings")
    n = _n_chunks_proxy(txt)
    seed = int.from_bytes(hashlib.sha256(txt.encode()).digest()[:4], "little")
    return torch.from_numpy(np.random.default_rng(seed).standard_normal((n, 77, 768)).astype(np.float32)).to(dev).half()


def encode_prompt_pair(positive, negative, dev, enc_dir=None):
    """-> [2, 77*k, 768] f16 = cat([uncond, cond]) with the chunk-tiling rule of generate.py:122-133."""
    c, uc = _inner(positive, dev, enc_dir), _inner(negative, dev, enc_dir)
    k = max(len(c), len(uc))
    c = torch.cat([c] * math.ceil(k / len(c)))[:k].reshape(1, -1, 768)
    uc = torch.cat([uc] * math.ceil(k / len(uc)))[:k].reshape(1, -1, 768)
    return torch.cat([uc, c]).contiguous()

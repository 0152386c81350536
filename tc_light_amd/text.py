"""Prompt embeddings (reference: generate.py:98-135 encode_prompt_inner / encode_prompt_pair).

Host plumbing, outside the two hot paths.  `encode_prompt_inner(txt, tokenizer, text_encoder, device)` is the reference's chunked
scheme on any tokenizer / encoder with the HF CLIP interface (`tokenizer(txt, truncation=False, add_special_tokens=False)["input_ids"]`,
`.model_max_length`, `.bos_token_id`, `.eos_token_id`; `text_encoder(ids).last_hidden_state`): untruncated tokens -> chunks of
model_max_length-2 wrapped in BOS/EOS, padded with EOS to model_max_length -> last_hidden_state per chunk.  `encode_prompt_pair` tiles the
shorter side, concatenates the chunks along the sequence and returns cat([uncond, cond]) (generate.py:553-555).
`load_text_encoder(dir)` loads CLIP ViT-L/14 with `transformers` from a local directory; a missing directory raises unless random
stand-ins were explicitly allowed (model_utils.allow_random), in which case deterministic text-seeded embeddings of the right shape are used.
"""
import hashlib
import math
import os
import warnings

import numpy as np
import torch

_ENC = {}


def encode_prompt_inner(txt, tokenizer, text_encoder, device):
    """A4 (generate.py:98-114): the untruncated token stream laid out as an id matrix [n_chunks, model_max_length] -- column 0 is BOS, the
    next model_max_length-2 columns carry the stream row after row, every other cell (the closing column and the ragged end of the
    last row) is EOS -- and encoded in one batch -> last_hidden_state [n_chunks, model_max_length, hidden]."""
    width = int(tokenizer.model_max_length)
    body = width - 2
    stream = torch.as_tensor(list(tokenizer(txt, truncation=False, add_special_tokens=False)["input_ids"]), dtype=torch.int64)
    rows = -(-int(stream.numel()) // body)
    if rows == 0:
        raise ValueError("empty prompt: the reference's chunking yields no chunk to encode (generate.py:108-112)")
    cells = torch.full((rows * body,), int(tokenizer.eos_token_id), dtype=torch.int64)
    cells[:stream.numel()] = stream
    ids = torch.full((rows, width), int(tokenizer.eos_token_id), dtype=torch.int64)
    ids[:, 0] = int(tokenizer.bos_token_id)
    ids[:, 1:1 + body] = cells.view(rows, body)
    with torch.no_grad():
        return text_encoder(ids.to(device)).last_hidden_state


def tile_and_concat(c, uc):
    """A4 (generate.py:116-135): both embeddings get the chunk count of the longer prompt -- the shorter one cycles through its chunks --
    and the chunks become one sequence.  c, uc: [n_chunks, L, D] -> ([1, k*L, D], [1, k*L, D])."""
    k = max(c.shape[0], uc.shape[0])

    def as_sequence(e):
        cycled = e.repeat(-(-k // e.shape[0]), 1, 1).narrow(0, 0, k)
        return cycled.reshape(1, k * e.shape[1], e.shape[2])

    return as_sequence(c), as_sequence(uc)


def load_text_encoder(enc_dir, dev):
    if enc_dir not in _ENC:
        from transformers import CLIPTextModel, CLIPTokenizer
        _ENC[enc_dir] = (CLIPTokenizer.from_pretrained(enc_dir), CLIPTextModel.from_pretrained(enc_dir).to(dev).half().eval())
    return _ENC[enc_dir]


def _n_chunks_proxy(txt):
    return max(1, math.ceil(len(txt.replace(",", " , ").replace(".", " . ").split()) * 1.3 / 75))


def _stand_in(txt, dev):
    n = _n_chunks_proxy(txt)
    seed = int.from_bytes(hashlib.sha256(txt.encode()).digest()[:4], "little")
    return torch.from_numpy(np.random.default_rng(seed).standard_normal((n, 77, 768)).astype(np.float32)).to(dev).half()


def encode_prompt_pair(positive, negative, dev, enc_dir=None, allow_random=False, tokenizer=None, text_encoder=None):
    """-> [2, L*k, 768] f16 = cat([uncond, cond]) (generate.py:553-555)."""
    if tokenizer is None and enc_dir and os.path.isdir(enc_dir):
        tokenizer, text_encoder = load_text_encoder(enc_dir, dev)
    if tokenizer is not None:
        c = encode_prompt_inner(positive, tokenizer, text_encoder, dev)
        uc = encode_prompt_inner(negative, tokenizer, text_encoder, dev)
    else:
        if not allow_random:
            raise FileNotFoundError(f"CLIP text encoder directory {enc_dir!r} not found (models.text_encoder); set models.allow_random / "
                                    "TCL_ALLOW_RANDOM_WEIGHTS=1 for deterministic stand-in embeddings")
        warnings.warn("CLIP text encoder not found -> deterministic text-seeded stand-in embeddings (allow_random)")
        c, uc = _stand_in(positive, dev), _stand_in(negative, dev)
    c, uc = tile_and_concat(c, uc)
    return torch.cat([uc, c]).to(torch.float16).contiguous()

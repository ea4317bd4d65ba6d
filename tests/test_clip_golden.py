"""CLIP towers (lvdm_amd/clip.py; condition.py:174-372) against tests/golden/clip_ref.npz: the reference's own
`encode_with_transformer` / `encode_with_vision_transformer` driving torch.nn.MultiheadAttention blocks with open_clip's parameter
tree (tests/golden/make_golden_clip.py).  CPU: fp32, tight; `-m gpu`: the fp16 module on the device."""
import os

import numpy as np
import pytest
import torch

from fill_by_name import fill_by_name

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "clip_ref.npz"), allow_pickle=False)
TEXT = dict(embed_dim=32, text_width=64, text_layers=3, text_heads=4, vocab=96, ctx=77)
VIS = dict(embed_dim=32, vision_cfg=dict(width=96, layers=3, heads=4, patch=8, image=32))


def _towers(device, dtype):
    from lvdm_amd.clip import FrozenOpenCLIPEmbedder, FrozenOpenCLIPImageEmbedderV2
    out = {}
    for layer in ("last", "penultimate"):
        t = FrozenOpenCLIPEmbedder(layer=layer, model_cfg=dict(TEXT), device=str(device))
        fill_by_name(t.model, std=0.08)
        out[layer] = t.to(device=device, dtype=dtype)
    v = FrozenOpenCLIPImageEmbedderV2(model_cfg=dict(VIS), device=str(device))
    fill_by_name(v.model, std=0.08)
    v.preprocess = lambda x: x                       # (the golden takes the kornia resize out on both sides)
    out["vis"] = v.to(device=device, dtype=dtype)
    return out


def test_state_dict_keys_are_open_clips():
    t = _towers("cpu", torch.float32)
    assert sorted(t["last"].model.state_dict().keys()) == list(G["text_keys"])
    # (open_clip's model keeps its text-side embeddings after `del model.transformer`, so does ours; the golden's stand-in has the
    #  visual tower only)
    assert sorted(k for k in t["vis"].model.state_dict().keys() if k.startswith("visual.")) == list(G["vis_keys"])


def test_towers_match_the_reference_call_sequence_fp32():
    t = _towers("cpu", torch.float32)
    tokens, img = torch.tensor(G["text_tokens"]), torch.tensor(G["vis_image"])
    with torch.no_grad():
        for layer in ("last", "penultimate"):
            np.testing.assert_allclose(t[layer].encode_with_transformer(tokens).numpy(), G[f"text_{layer}"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(t["vis"].encode_with_vision_transformer(img).numpy(), G["vis_tokens"], rtol=0, atol=2e-5)
    assert float(np.abs(G["text_last"] - G["text_penultimate"]).max()) > 1e-2      # the layer switch is visible


@pytest.mark.gpu
def test_towers_in_fp16_on_the_device():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm device")
    t = _towers("cuda:0", torch.float16)
    tokens, img = torch.tensor(G["text_tokens"], device="cuda:0"), torch.tensor(G["vis_image"], device="cuda:0").half()
    with torch.no_grad():
        for layer in ("last", "penultimate"):
            ref = torch.tensor(G[f"text_{layer}"])
            got = t[layer].encode_with_transformer(tokens).float().cpu()
            assert float((got - ref).abs().max() / ref.abs().max()) < 6e-3
        ref = torch.tensor(G["vis_tokens"])
        got = t["vis"].encode_with_vision_transformer(img).float().cpu()
        assert float((got - ref).abs().max() / ref.abs().max()) < 6e-3

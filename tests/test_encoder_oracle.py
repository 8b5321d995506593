"""CPU: the encoder oracle against the installed `transformers` package, and the encoder modules'
parameter names / checkpoint import (no kernel runs here)."""
import os

import pytest
import torch

os.environ.setdefault("HF_HUB_OFFLINE", "1")
transformers = pytest.importorskip("transformers")

from oracle import encoder_oracle as eo  # noqa: E402
from vqa_amd import encoders  # noqa: E402


def _vit_keys_to_pinned(sd):
    """Key names of the installed release -> those of transformers 4.52 checkpoints."""
    out = {}
    for k, v in sd.items():
        k = k.replace("layers.", "encoder.layer.") if k.startswith("layers.") else k
        k = (k.replace(".attention.q_proj.", ".attention.attention.query.")
              .replace(".attention.k_proj.", ".attention.attention.key.")
              .replace(".attention.v_proj.", ".attention.attention.value.")
              .replace(".attention.o_proj.", ".attention.output.dense.")
              .replace(".mlp.fc1.", ".intermediate.dense.")
              .replace(".mlp.fc2.", ".output.dense."))
        out[k] = v
    return out


def _tiny_vit(seed=0):
    torch.manual_seed(seed)
    cfg = transformers.ViTConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                 image_size=48, patch_size=16)
    m = transformers.ViTModel(cfg, add_pooling_layer=True).eval()
    with torch.no_grad():  # the package zero-initialises some tensors; make every one count
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * 0.05)
    return cfg, m


def _tiny_bert(seed=0):
    torch.manual_seed(seed)
    cfg = transformers.BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                  vocab_size=97, max_position_embeddings=40)
    m = transformers.BertModel(cfg).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * 0.05)
    return cfg, m


def test_vit_oracle_reproduces_transformers():
    cfg, m = _tiny_vit()
    sd = _vit_keys_to_pinned(m.state_dict())
    x = torch.randn(3, 3, 48, 48)
    with torch.no_grad():
        ref = m(pixel_values=x)
        h, pooled = eo.vit_forward(sd, x, heads=2, patch=16, eps=cfg.layer_norm_eps)
    torch.testing.assert_close(h, ref.last_hidden_state, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(pooled, ref.pooler_output, rtol=1e-5, atol=1e-6)


def test_bert_oracle_reproduces_transformers_with_padding():
    cfg, m = _tiny_bert()
    sd = m.state_dict()
    ids = torch.randint(1, 97, (4, 11))
    mask = torch.ones(4, 11, dtype=torch.long)
    mask[1, 7:] = 0
    mask[3, 3:] = 0
    with torch.no_grad():
        ref = m(input_ids=ids, attention_mask=mask)
        h, pooled = eo.bert_forward(sd, ids, mask, heads=2, eps=cfg.layer_norm_eps)
    torch.testing.assert_close(h, ref.last_hidden_state, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(pooled, ref.pooler_output, rtol=1e-5, atol=1e-6)


def test_roberta_oracle_reproduces_transformers_with_padding():
    torch.manual_seed(2)
    cfg = transformers.RobertaConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                     vocab_size=97, max_position_embeddings=40, type_vocab_size=1, pad_token_id=1,
                                     layer_norm_eps=1e-5)
    m = transformers.RobertaModel(cfg).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * 0.05)
    ids = torch.randint(2, 97, (4, 11))
    mask = torch.ones(4, 11, dtype=torch.long)
    mask[1, 7:] = 0
    ids[1, 7:] = 1
    with torch.no_grad():
        ref = m(input_ids=ids, attention_mask=mask)
        h, pooled = eo.bert_forward(m.state_dict(), ids, mask, heads=2, eps=1e-5, roberta_pad=1)
    torch.testing.assert_close(h, ref.last_hidden_state, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(pooled, ref.pooler_output, rtol=1e-5, atol=1e-6)
    mine = encoders.BertModel(encoders.roberta_config(hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                                                      intermediate_size=256, vocab_size=97, max_position_embeddings=40))
    want = {k: tuple(v.shape) for k, v in m.state_dict().items() if "position_ids" not in k and "token_type_ids" not in k}
    assert {k: tuple(v.shape) for k, v in mine.state_dict().items()} == want
    wrapper = encoders.RoBerta(hidden_dim=64, config=encoders.roberta_config(
        hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256, vocab_size=97,
        max_position_embeddings=40))
    assert wrapper.output_dim == 64 and wrapper.bert.config.roberta_positions


def test_module_parameter_names_are_the_checkpoint_names():
    cfg, m = _tiny_vit()
    want = {k: tuple(v.shape) for k, v in _vit_keys_to_pinned(m.state_dict()).items()}
    mine = encoders.ViTModel(encoders.ViTConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                                                intermediate_size=256, image_size=48))
    assert {k: tuple(v.shape) for k, v in mine.state_dict().items()} == want
    cfg, m = _tiny_bert()
    want = {k: tuple(v.shape) for k, v in m.state_dict().items() if "position_ids" not in k}
    mine = encoders.BertModel(encoders.BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                                                  intermediate_size=256, vocab_size=97, max_position_embeddings=40))
    assert {k: tuple(v.shape) for k, v in mine.state_dict().items()} == want


def test_wrappers_follow_the_reference_contract():
    """model/text_encoder.py:18-31 and model/vision_encoder.py:91-106: attribute names, projections, output_dim."""
    t = encoders.Bert(hidden_dim=512, config=encoders.BertConfig(hidden_size=128, num_hidden_layers=1,
                                                                 num_attention_heads=2, intermediate_size=256,
                                                                 vocab_size=50, max_position_embeddings=16))
    assert t.output_dim == 512 and t.seq_projection.out_features == 512 and t.global_projection.in_features == 128
    assert any(k.startswith("bert.encoder.layer.0.attention.self.query") for k in t.state_dict())
    v = encoders.ViT(out_features=512, config=encoders.ViTConfig(hidden_size=128, num_hidden_layers=1,
                                                                 num_attention_heads=2, intermediate_size=256,
                                                                 image_size=32))
    assert v.output_dim == 512 and v.seq_projection.out_features == 512


def test_local_checkpoint_import(tmp_path):
    from safetensors.torch import save_file
    cfg, m = _tiny_bert(seed=3)
    d = tmp_path / "bert-tiny"
    d.mkdir()
    # a task checkpoint: `bert.` prefix, a head the encoder does not have, old LayerNorm names
    sd = {}
    for k, v in m.state_dict().items():
        k = k.replace("embeddings.LayerNorm.weight", "embeddings.LayerNorm.gamma").replace(
            "embeddings.LayerNorm.bias", "embeddings.LayerNorm.beta")
        sd["bert." + k] = v.contiguous()
    sd["cls.predictions.bias"] = torch.zeros(97)
    save_file(sd, str(d / "model.safetensors"))
    (d / "config.json").write_text(cfg.to_json_string())
    enc = encoders.Bert(model_name=str(d), hidden_dim=64)
    assert enc.bert.config.hidden_size == 128 and enc.bert.config.num_hidden_layers == 2
    for k, v in m.state_dict().items():
        if "position_ids" in k:
            continue
        assert torch.equal(enc.bert.state_dict()[k], v), k
    # a file that lacks a tensor of the model is refused
    del sd["bert.pooler.dense.weight"]
    save_file(sd, str(d / "model.safetensors"))
    with pytest.raises(KeyError):
        encoders.Bert(model_name=str(d), hidden_dim=64)


def test_encoders_refuse_cpu_tensors():
    enc = encoders.Bert(hidden_dim=64, config=encoders.BertConfig(hidden_size=128, num_hidden_layers=1,
                                                                  num_attention_heads=2, intermediate_size=256,
                                                                  vocab_size=50, max_position_embeddings=16))
    with pytest.raises(RuntimeError, match="GPU only"):
        enc(torch.randint(1, 50, (2, 5)), torch.ones(2, 5, dtype=torch.long))


def test_local_checkpoint_import_vit_torch_file(tmp_path):
    """A `pytorch_model.bin` task checkpoint (`vit.` prefix, a classifier head) read with weights_only=True."""
    cfg, m = _tiny_vit(seed=5)
    d = tmp_path / "vit-tiny"
    d.mkdir()
    sd = {"vit." + k: v.clone() for k, v in _vit_keys_to_pinned(m.state_dict()).items()}
    sd["classifier.weight"] = torch.zeros(10, 128)
    torch.save(sd, str(d / "pytorch_model.bin"))
    (d / "config.json").write_text(cfg.to_json_string())
    enc = encoders.ViT(model_name=str(d), out_features=64)
    assert enc.vit.config.image_size == 48 and enc.vit.config.num_hidden_layers == 2
    for k, v in _vit_keys_to_pinned(m.state_dict()).items():
        assert torch.equal(enc.vit.state_dict()[k], v), k

"""vqaf_embed_fwd / vqaf_embed_bwd (word + position + token-type rows per token, and the three table gradients)
against torch index arithmetic in f64.  Tolerance: f32 tables are summed in f32 (1e-6 relative); bf16 outputs are
rounded once (2^-8 relative)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


@pytest.mark.parametrize("pdtype,odtype", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                           (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("B,L,V,E", [(64, 30, 30522, 768), (3, 7, 50, 64), (1, 1, 9, 8)])
def test_embedding_sum_and_table_gradients(pdtype, odtype, B, L, V, E):
    from vqa_amd.encoders import _EmbeddingSum
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(B * 131 + L)
    word = torch.randn(V, E, generator=g).to(pdtype).to(dev).requires_grad_()
    pos = torch.randn(512, E, generator=g).to(pdtype).to(dev).requires_grad_()
    typ = torch.randn(2, E, generator=g).to(pdtype).to(dev).requires_grad_()
    ids = torch.randint(0, V, (B, L), generator=g)
    ids[:, 0] = 5                      # one id carried by every sample (a [CLS]-like token)
    ids[0, -1] = 0                     # a padding token
    if B > 1:
        ids[1] = ids[0]                # duplicates across samples
    ids = ids.to(dev)
    pos_ids = torch.arange(L, device=dev).repeat(B, 1)
    type_ids = torch.zeros_like(ids)
    if B > 2:
        type_ids[2] = 1
    out = _EmbeddingSum.apply(word, pos, typ, ids, pos_ids, type_ids, odtype, 0)
    want = word.double()[ids] + pos.double()[pos_ids] + typ.double()[type_ids]
    rel = 2.0 ** -8 if odtype == torch.bfloat16 else 1e-6
    assert out.dtype == odtype and out.shape == (B, L, E)
    assert (out.double() - want).abs().max().item() <= rel * want.abs().max().item() + 1e-6

    cot = torch.randn(B, L, E, generator=g).to(odtype).to(dev)
    out.backward(cot)
    c64 = cot.double().reshape(-1, E)
    gw = torch.zeros(V, E, dtype=torch.float64, device=dev).index_add_(0, ids.reshape(-1), c64)
    gw[0] = 0                          # padding_idx row
    gp = torch.zeros(512, E, dtype=torch.float64, device=dev).index_add_(0, pos_ids.reshape(-1), c64)
    gt = torch.zeros(2, E, dtype=torch.float64, device=dev).index_add_(0, type_ids.reshape(-1), c64)
    relg = 2.0 ** -8 if pdtype == torch.bfloat16 else 1e-5
    for got, ref in ((word.grad, gw), (pos.grad, gp), (typ.grad, gt)):
        assert got.dtype == pdtype and got.shape == ref.shape
        assert (got.double() - ref).abs().max().item() <= relg * ref.abs().max().item() + 1e-5
    # fixed summation order: the same bits on a second pass
    first = [t.grad.clone() for t in (word, pos, typ)]
    for t in (word, pos, typ):
        t.grad = None
    _EmbeddingSum.apply(word, pos, typ, ids, pos_ids, type_ids, odtype, 0).backward(cot)
    for a, t in zip(first, (word, pos, typ)):
        assert torch.equal(a, t.grad)


def test_bert_embeddings_take_the_one_launch_path_and_match_the_torch_path():
    from vqa_amd import encoders
    dev = _dev()
    torch.manual_seed(0)
    cfg = encoders.BertConfig(num_hidden_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    emb = encoders._BertEmbeddings(cfg).to(dev)
    ids = torch.randint(1, cfg.vocab_size, (8, 30), device=dev)
    ids[:, -3:] = cfg.pad_token_id
    y = emb(ids, torch.float32)
    y.sum().backward()
    got = [p.grad.clone() for p in (emb.word_embeddings.weight, emb.position_embeddings.weight,
                                    emb.token_type_embeddings.weight)]
    emb.zero_grad()
    # the torch path of the same module: what it computes when the library path does not apply
    pos = emb.position_embeddings.weight[:30] + emb.token_type_embeddings.weight[0]
    x = emb.word_embeddings(ids) + pos
    ref, _ = encoders.add_layer_norm(x, None, emb.LayerNorm.weight, emb.LayerNorm.bias, emb.eps)
    torch.testing.assert_close(y, ref, rtol=1e-5, atol=1e-5)
    ref.sum().backward()
    want = [emb.word_embeddings.weight.grad, emb.position_embeddings.weight.grad,
            emb.token_type_embeddings.weight.grad]
    for a, b in zip(got, want):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)

"""The row-sharded hot path on the PRODUCT backend (dist.HipBackend: HIP kernels incl. the owned-row gather / scatter
pair) at world size 1 against the single-process oracle on the G8 problem: loss and every gradient, fused and composed
paths. (World sizes 2 and 3 run on CPU with the oracle backend in test_dist_cpu.py; N > 1 on GPUs is the driver's.)"""
import os
import sys

import pytest
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solo_group():
    import test_dist_cpu as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(T._free_port())
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", rank=0, world_size=1)
    yield
    if created:
        dist.destroy_process_group()


@pytest.mark.parametrize("modal,fused", [("full", True), ("empty", True), ("empty_shortcut", True), ("full", False)])
def test_sharded_step_on_hip_backend_world1_equals_oracle(solo_group, modal, fused):
    import mmssl_oracle as O
    import test_dist_cpu as T
    from mmssl_amd import dist as md
    dev = torch.device("cuda")
    fx, d, raw, U, I, state, users, pos, neg, img_raw, txt_raw = T._global_problem(modal)
    ush, ish = md.RowShard(U, 1, 0), md.RowShard(I, 1, 0)
    bk = md.HipBackend()
    cfg = O.Cfg(drop_rate=0.0, batch_size=48, n_ui_layers=2)

    def local_pair(m):
        ui, iu = O.csr_norm(m, True), O.csr_norm(m.T, True)
        return bk.make_graph(md.shard_graph(ui, ush, ish)), bk.make_graph(md.shard_graph(iu, ish, ush))
    graphs = local_pair(raw) + local_pair(img_raw) + local_pair(txt_raw)
    d, state, k_txt = T._pad_text_to_slices(d, state)
    model = md.ShardedMMSSL(bk, cfg, ush, ish, state, d["image_feat"], d["text_feat"]).to(dev).train()
    step = md.ShardedHotPathStep(model, graphs, 48, I, modal_empty=(modal == "empty_shortcut"), optimizer=False,
                                 fused=fused)
    step.set_batch(torch.stack([users, pos, neg]).to(dev))           # packed form: one copy
    total = step.backward()
    torch.cuda.synchronize()
    ref_loss, P = T._reference(modal)
    assert abs(float(total) - ref_loss) <= 2e-5 * abs(ref_loss), (float(total), ref_loss)

    def rel(a, b):
        return float((a.cpu() - b).abs().max() / (b.abs().max() + 1e-30))
    assert model.last_fused == fused                 # the packed node really ran
    g = {n: p.grad for n, p in model.named_parameters()}
    assert float(g["txt_w"][:, k_txt:].abs().max()) == 0.0
    g["txt_w"] = g["txt_w"][:, :k_txt]
    for name, key in (("img_w", "image_trans.weight"), ("img_b", "image_trans.bias"), ("txt_w", "text_trans.weight"),
                      ("txt_b", "text_trans.bias"), ("E_u", "user_id_embedding.weight"), ("E_i", "item_id_embedding.weight")):
        k = P[key].grad.shape[0]
        assert rel(g[name][:k], P[key].grad) < 1e-4, name
    if modal == "full":
        assert rel(g["w_cat"], P["weight_dict.w_self_attention_cat"].grad) < 1e-4

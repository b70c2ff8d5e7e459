"""Whole-loop timing of mmssl_amd.main.Trainer (the reference's training loop: sampler -> D step ->
G step per batch, evaluation per epoch) on a synthetic dataset of a named shape written in the
reference's on-disk format. Prints a per-phase breakdown (ms per batch, CUDA-synchronised).

    python tools/trainer_bench.py [--workload baby|tiktok] [--batches 12]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="baby")
    ap.add_argument("--batches", type=int, default=12)
    ap.add_argument("--eval-users", type=int, default=1 << 30, help="validation users per test() call (default: all)")
    a = ap.parse_args()
    import synth_data
    from mmssl_amd import synth
    from mmssl_amd.config import configure
    from mmssl_amd.utility import batch_test
    from mmssl_amd import main as M
    U, I, E, dv, dt = synth.SHAPES[a.workload]
    root = tempfile.mkdtemp(prefix="mmssl_bench_")
    t0 = time.time()
    synth_data.write_dataset(root, a.workload, U, I, E, dv, dt, seed=1)
    t_data = time.time() - t0
    configure(["--data_path", root + "/", "--dataset", a.workload, "--weight_size", "[64,64,64]", "--verbose", "0"])
    M.set_seed(2022)
    dg = batch_test.init_data()
    tr = M.Trainer({"n_users": dg.n_users, "n_items": dg.n_items})
    sync = torch.cuda.synchronize
    # phase breakdown (synchronised after each phase) ...
    acc = {"sample": 0.0, "d_step": 0.0, "g_step": 0.0}
    n = 0
    for idx in range(a.batches):
        tr.model.train()
        t = time.perf_counter()
        users, pos, neg = dg.sample()
        t1 = time.perf_counter()
        tr._batch_idx(users, pos, neg)
        tr._discriminator_step(users)
        sync()
        t2 = time.perf_counter()
        tr._generator_step(idx, users, pos, neg)
        sync()
        t3 = time.perf_counter()
        if idx >= 2:                     # first two batches build caches / modal graphs
            acc["sample"] += t1 - t
            acc["d_step"] += t2 - t1
            acc["g_step"] += t3 - t2
            n += 1
    out = {k: round(v / n * 1e3, 2) for k, v in acc.items()}
    out["phases_sum_ms"] = round(sum(acc.values()) / n * 1e3, 2)
    # The D step split into its IN-SCOPE part (the model forward that feeds it and the three u_sim_calculation
    # calls, rows a-5 and f-1) and the OUT-OF-SCOPE GAN arithmetic around them (Discriminator MLP forward / backward,
    # Gumbel noise, gradient penalty, D's optimiser: SURVEY 2 rows 3 and 7), each piece timed on its own, synchronised
    users, pos, neg = dg.sample()
    tr._batch_idx(users, pos, neg)

    def timed(fn, reps=5):
        fn()
        sync()
        t = time.perf_counter()
        for _ in range(reps):
            r = fn()
        sync()
        return (time.perf_counter() - t) / reps * 1e3, r
    with torch.no_grad():
        t_fwd, outs = timed(lambda: tr.model(*tr._graphs()))
        ua, ia, img_item, txt_item, img_user, txt_user = outs[:6]
        t_usim, _ = timed(lambda: (tr.u_sim_calculation(users, ua, ia), tr.u_sim_calculation(users, img_user, img_item),
                                   tr.u_sim_calculation(users, txt_user, txt_item)))
    t_d, _ = timed(lambda: tr._discriminator_step(users))
    out["d_step_parts_ms"] = {"model_forward_in_scope": round(t_fwd, 3), "u_sim_x3_in_scope": round(t_usim, 3),
                              "gan_ops_out_of_scope": round(max(t_d - t_fwd - t_usim, 0.0), 3), "d_step_total": round(t_d, 3)}
    t_g, _ = timed(lambda: tr._generator_step(a.batches + 50, users, pos, neg))
    out["g_step_total_ms"] = round(t_g, 3)
    out["in_scope_share_of_d_step"] = round((t_fwd + t_usim) / max(t_d, 1e-9), 4)
    # ... and the loop exactly as Trainer.train() runs it (next batch sampled while the device works)
    # (Trainer.train_batch: op by op with MMSSL_TRAINER_GRAPH=0, then on the two captured hot-path segments)
    base = a.batches
    for tag, flag in (("batch_total_ms_eager", "0"), ("batch_total_ms_captured", "1")):
        os.environ["MMSSL_TRAINER_GRAPH"] = flag
        nxt = dg.sample()
        for idx in range(3):                              # reach / capture the steady state outside the timed loop
            users, pos, neg = nxt
            tr._batch_idx(users, pos, neg)
            float(tr.train_batch(base + idx, users, pos, neg)[0])
            nxt = dg.sample()
        sync()
        t0 = time.perf_counter()
        for idx in range(a.batches):
            users, pos, neg = nxt
            tr._batch_idx(users, pos, neg)
            bl = tr.train_batch(base + 3 + idx, users, pos, neg)[0]
            nxt = dg.sample()
            float(bl)
        out[tag] = round((time.perf_counter() - t0) / a.batches * 1e3, 2)
        base += a.batches + 3
    out["captured_path_used"] = getattr(tr, "_split", None) not in (None, False)
    os.environ.pop("MMSSL_TRAINER_GRAPH", None)
    users = list(dg.val_set.keys())[:a.eval_users]
    sync()
    t = time.perf_counter()
    ret = tr.test(users, is_val=True)          # first call: builds and uploads the train / validation CSRs once per dataset
    sync()
    out["eval_first_call_ms"] = round((time.perf_counter() - t) * 1e3, 2)
    t = time.perf_counter()
    ret = tr.test(users, is_val=True)          # what every later epoch pays: eval-mode forward + scores + top-K + metrics
    sync()
    out["eval_users"] = len(users)
    out["eval_ms_per_1k_users"] = round((time.perf_counter() - t) * 1e3 / max(len(users), 1) * 1000, 3)
    out["recall@20"] = float(ret["recall"][1])
    out["workload"] = a.workload
    out["dataset_write_s"] = round(t_data, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

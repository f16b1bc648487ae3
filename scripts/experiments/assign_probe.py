"""Matcher probe (GPU): the cost matrices of the bench step, and the time of the assignment kernel alone.

    python scripts/experiments/assign_probe.py dump OUT.npz     run 4 bench-shaped train steps, save the [48,100,99] cost
                                                               tensor + targets of the last one
    python scripts/experiments/assign_probe.py time IN.npz      time detr_hip_assign_f32 on that tensor and on iid-uniform
                                                               matrices with n = 99 / 50 / 20 / 7 targets
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "detr-tensorflow_amd"))
sys.path.insert(0, ROOT)


def time_assign(hip, cost, tb, B, reps=20):
    P, Q, ldc = cost.shape
    R = tb.shape[1]
    dev = cost.device
    tfp = torch.empty(P, Q, dtype=torch.int32, device=dev)
    pft = torch.empty(P, ldc, dtype=torch.int32, device=dev)
    st = torch.empty(P, dtype=torch.int32, device=dev)

    def run():
        hip.call("detr_hip_assign_f32", cost.data_ptr(), P, Q, ldc, tb.data_ptr(), B, R, tfp.data_ptr(), pft.data_ptr(),
                 st.data_ptr())
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, pft.cpu().numpy(), st.cpu().numpy()


def main():
    mode, path = sys.argv[1], sys.argv[2]
    dev = torch.device("cuda:0")
    from detr_tf import _hip as hip
    if mode == "dump":
        import bench
        from detr_tf import training
        from detr_tf.loss.loss import SetLoss
        from detr_tf.networks.detr import get_detr_model
        from detr_tf.optimizers import setup_optimizers
        from detr_tf.training_config import TrainingConfig
        cfg = TrainingConfig()
        cfg.background_class = 91
        cfg.batch_size = 8
        cfg.target_batch = None
        cfg.train_backbone = cfg.train_transformers = cfg.train_nlayers = True
        m = get_detr_model(cfg, include_top=True, device=str(dev), seed=0, dropout=0.1, precision="bf16")
        o = setup_optimizers(m, cfg)
        rng = np.random.default_rng(1234)
        images = torch.from_numpy(rng.normal(size=(8, 800, 1333, 3)).astype(np.float32)).to(dev)
        tb, tc = bench.make_targets(8, np.random.default_rng(1235))
        tb, tc = torch.from_numpy(tb).to(dev), torch.from_numpy(tc).to(dev)
        out = {}
        for i in range(12):
            training.train_step(m, images, tb, tc, o, cfg, i)
            if i in (0, 3, 11):
                sl = next(iter(SetLoss._cache.values()))
                out[f"cost{i}"] = sl.matcher.cost.cpu().numpy()
                out[f"pft{i}"] = sl.matcher.pred_for_tgt.cpu().numpy()
        out["t_bbox"] = tb.cpu().numpy()
        np.savez_compressed(path, **out)
        print("saved", path, {k: v.shape for k, v in out.items()})
        return
    d = np.load(path)
    tb = torch.from_numpy(d["t_bbox"]).to(dev)
    for k in sorted(k for k in d.files if k.startswith("cost")):
        cost = torch.from_numpy(d[k]).to(dev)
        us, pft, st = time_assign(hip, cost, tb, tb.shape[0])
        print(f"bench {k}: {us:8.1f} us  status {st.max()}")
    rng = np.random.default_rng(0)
    for n in (99, 50, 20, 7):
        t = np.zeros((8, 100, 4), np.float32)
        t[:, 0, 0] = n
        cost = torch.from_numpy(rng.uniform(0, 1, (48, 100, 99)).astype(np.float32)).to(dev)
        us, _, _ = time_assign(hip, cost, torch.from_numpy(t).to(dev), 8)
        print(f"iid uniform n={n:3d}: {us:8.1f} us")
    # one problem alone (no other wave on the chip): latency of a single 99-target problem
    t = np.zeros((1, 100, 4), np.float32)
    t[:, 0, 0] = 99
    cost = torch.from_numpy(d["cost3"][47:48]).to(dev)
    us, pft, _ = time_assign(hip, cost, torch.from_numpy(t).to(dev), 1)
    print(f"bench cost3[47] alone: {us:8.1f} us   (probe builds: s_memtime segments {pft[0, :7].tolist()})")


if __name__ == "__main__":
    main()

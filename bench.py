#!/usr/bin/env python3
"""Decode-GEMV benchmark for the VPTQ hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--hidden 8192] [--mode single|grouped]

Workload (BASELINE.json configs[1] at the hidden size the north-star target is
quoted on): one VQuantLinear H x H, vector_len 8, k = 256 + 256 residual
centroids (2 bits / weight), enable_norm, batch = 1, seq = 1, fp16.
A *step* is one decode token through a RING of R distinct layers
(R * packed-index bytes >= 512 MiB, so neither the 32 MiB of L2 nor the 256 MiB
Infinity Cache can hold the weights between uses): R fused dequant+GEMV
launches through the C ABI, captured once in a hipGraph and replayed.
  --mode single  : one launch per layer (the reference's operator granularity)
  --mode grouped : layers launched 4 at a time with vptq_quant_gemv_grouped
                   (q/k/v/o-style fusion of independent projections)
  --mode tp      : BASELINE config #5 flavour: every layer is cut into N slices of its output
                   rows (vptq_amd.utils.shard.shard_out_features), one per rank, and the slices
                   are re-assembled with an RCCL all-gather per layer; strong scaling (total
                   work fixed), expected NOT to scale at batch 1
Inputs and weights are resident in HBM before the timed region.  With N > 1
(torchrun, one rank per GPU) every rank owns its own ring: independent layers, no
data-path collective, weak scaling; the time is the max over ranks.

Prints ONE JSON line; see README / DESIGN.md §6 for the fields.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)


def alg_bytes(H):
    """Algorithmic bytes of one H x H launch (DESIGN.md §5 / SURVEY.md §8d):
    packed indices + both codebooks + x + scale + bias + y."""
    return (H // 8) * (H * 16 // 32) * 4 + 2 * 256 * 8 * 2 + 2 * H + 4 * H + 2 * H


def shard_ring(total_layers, rank, world):
    """Layer-parallel sharding: independent VQuantLinear layers are dealt round-robin to
    ranks; no rank ever needs another rank's layer (no data-path collective)."""
    return list(range(rank, total_layers, world))


def reduce_times(wall_s, event_ms, dist_mod=None, device="cpu"):
    """max over ranks of (wall seconds, HIP-event milliseconds)."""
    if dist_mod is None:
        return wall_s, event_ms
    tt = torch.tensor([wall_s, event_ms], device=device, dtype=torch.float64)
    dist_mod.all_reduce(tt, op=dist_mod.ReduceOp.MAX)
    return tuple(tt.tolist())


def job_throughput_gbps(world, bytes_per_launch_layer, layers_per_rank, steps, wall_s):
    """Whole-job GB/s: every rank streams its own ring `steps` times in `wall_s`."""
    return world * bytes_per_launch_layer * layers_per_rank * steps / wall_s / 1e9


def make_ring(H, R, dev, seed):
    import vptq_amd
    g = torch.Generator(device=dev).manual_seed(seed)
    layers = []
    for _ in range(R):
        m = vptq_amd.VQuantLinear(
            H, H, vector_lens=[-1, 8], num_centroids=[-1, 256], num_res_centroids=[-1, 256],
            group_num=1, group_size=H, outlier_size=0, indices_as_float=False, enable_norm=True,
            enable_perm=False, is_indice_packed=True, bias=False, dtype=torch.float16,
            device=dev, enable_proxy_error=False)
        m.indices.data = torch.randint(-2**31, 2**31 - 1, m.indices.shape, generator=g,
                                       device=dev, dtype=torch.int64).to(torch.int32)
        m.centroids.weight.data = (torch.randn(m.centroids.weight.shape, generator=g, device=dev) * 0.02).half()
        m.res_centroids.weight.data = (torch.randn(m.res_centroids.weight.shape, generator=g, device=dev) * 0.005).half()
        m.weight_scale.data = (1 + 0.1 * torch.randn(H, generator=g, device=dev)).half()
        m.weight_bias.data = (0.01 * torch.randn(H, generator=g, device=dev)).half()
        layers.append(m.eval())
    return layers


def cpu_baseline(layer, x, y_gpu, H):
    """The reference's CPU algorithm (dequant to dense W, then linear), as restated by
    the C oracle (oracle/vptq_oracle.c), timed on this box's host cores.  Also returns
    the parity error of the GPU result against it."""
    from oracle import c_oracle as co
    from oracle import vptq_oracle as vo
    if not co.available():
        return None, None
    L = vo.LayerSpec(H, H, 8, 256, 256, 1, H, dtype="f16")
    to_np = lambda t, dt: t.detach().cpu().contiguous().view(torch.int16).numpy().view(dt)  # noqa: E731
    L.indices = layer.indices.detach().cpu().numpy()
    L.centroids = to_np(layer.centroids.weight, np.uint16).reshape(1, 256, 8)
    L.res_centroids = to_np(layer.res_centroids.weight, np.uint16).reshape(1, 256, 8)
    L.weight_scale = to_np(layer.weight_scale, np.uint16)
    L.weight_bias = to_np(layer.weight_bias, np.uint16)
    xb = to_np(x, np.uint16)
    scratch = np.empty((H, H), dtype=np.uint16)
    cores = co.lib().vo_num_threads()
    y = co.forward(L, xb, scratch=scratch)          # warm-up (page-in)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        y = co.forward(L, xb, scratch=scratch)
        times.append(time.perf_counter() - t0)
    t = sorted(times)[1]
    yg = to_np(y_gpu, np.uint16).reshape(-1)
    a = vo.to_f32(yg, "f16").astype(np.float64)
    b = vo.to_f32(y.reshape(-1), "f16").astype(np.float64)
    rel = float(np.abs(a - b).max() / np.abs(b).max())
    base = {"value": alg_bytes(H) / t / 1e9, "unit": "GB/s", "cores": int(cores), "kind": "port",
            "sample": f"1 VQuantLinear {H}x{H} forward (dequant to dense W + linear, C oracle, "
                      f"OpenMP {cores} threads), median of 3, {t * 1e3:.0f} ms each"}
    return base, rel


def compute_floor(kname, H):
    """What bounds the kernel besides HBM (tools/ubench.hip, tools/ubench_lds.hip on MI355X):
    per launch of one HxH layer, with every CU busy."""
    n_idx = (H // 8) * H                       # index pairs = weight vectors of 8
    if kname.startswith("gemv_k256m"):
        # 2 ds_read_b128 gathers per index, 5.1 LDS cycles per wave-instruction per CU, 256 CUs
        cyc = n_idx * 2 / 64 / 256 * 5.1
        # and the SIMD issue slots next to it: 4 MFMA 4x4x4 (8 cycles each) + 2-4 v_perm_b32 per
        # index-wave, 128 index-waves per SIMD and 8192^2 layer, which the SIMD does not overlap
        return {"what": "LDS gather throughput: 2 x ds_read_b128 per index at 5.1 LDS cycles per "
                        "wave-instruction and CU (conflict-free lane-split 8-replica image); SIMD "
                        "issue (4 MFMA + 2-4 VALU per index) is the same order, see DESIGN.md 4.1",
                "us_per_launch_at_2.1GHz": cyc / 2100.0}
    instr = 14 if "fast" in kname else 22
    return {"what": f"VALU issue: {instr} instructions per index, 2.25 ns per wave-instruction per "
                    "SIMD at 4 waves/SIMD (1024 SIMDs)",
            "us_per_launch": n_idx * instr / (1024 * 64) * 2.25e-3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--hidden", type=int, default=8192)
    ap.add_argument("--ring", type=int, default=0)
    ap.add_argument("--mode", choices=["single", "grouped", "tp", "tp_row"], default="single",
                    help="tp: output rows of every layer split over the ranks, RCCL all-gather; "
                         "tp_row: input columns split (row-parallel, BASELINE config #5), RCCL "
                         "all-reduce of the partial sums")
    ap.add_argument("--group", type=int, default=4)
    ap.add_argument("--exact", action="store_true",
                    help="VPTQ_GEMV_EXACT: rebuild every weight with the reference CPU path's three "
                         "16-bit roundings (bit-identical weights) instead of the default folded "
                         "fp32 form (both are inside the 1e-3 parity bar, checked below)")
    ap.add_argument("--fast-math", action="store_true", help="accepted, no effect: the default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prefetch", action="store_true",
                    help="let layer i read layer i+1's indices ahead (chain_prefetch).  Measured: "
                         "+2 %% throughput but 2x L2-fabric fetch traffic, so it is off by default")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from vptq_amd import _backend as B
    from _gpu_util import module_desc
    lib = B.lib()
    H = a.hidden
    idx_bytes = (H // 8) * H * 2
    R = a.ring or max(2, (512 << 20) // idx_bytes)
    # tp: all ranks build the same layers, each keeps its slice of the output rows
    tp = a.mode in ("tp", "tp_row")
    layers = make_ring(H, R, dev, seed=1234 + (0 if tp else rank))
    if a.mode == "tp_row":
        from vptq_amd.utils.shard import shard_in_features
        layers = [shard_in_features(m, rank, world) for m in layers]
        torch.cuda.empty_cache()
    if a.mode == "tp":
        from vptq_amd.utils.shard import shard_out_features
        layers = [shard_out_features(m, rank, world) for m in layers]
        torch.cuda.empty_cache()
    x = torch.randn(1, 1, H, device=dev, dtype=torch.float16,
                    generator=torch.Generator(device=dev).manual_seed(7))
    if a.mode == "tp_row":   # every rank multiplies its slice of the input columns
        g0, g1 = layers[0].shard[1], layers[0].shard[2]
        x = x[..., g0:g1].contiguous()
    ys = [torch.empty(1, 1, layers[i].out_features, device=dev, dtype=torch.float16)
          for i in range(R)]
    y_full = torch.empty(H, device=dev, dtype=torch.float16) if a.mode == "tp" else None
    descs, keeps = [], []
    for i, m in enumerate(layers):
        # decode order is known: layer i warms L2 / Infinity Cache with layer i+1's indices
        nxt = layers[(i + 1) % R].indices if a.prefetch else None
        d, k = module_desc(m, prefetch=nxt)
        descs.append(d)
        keeps.append(k)
    flags = B.GEMV_EXACT if a.exact else 0
    kname = lib.vptq_quant_gemv_kernel_name(descs[0], 1, flags).decode()

    stream = torch.cuda.Stream(device=dev)
    if a.mode == "tp_row":
        launches_per_step = R

        def one_pass():
            sp = torch.cuda.current_stream().cuda_stream
            for i in range(R):
                rc = lib.vptq_quant_gemv(descs[i], x.data_ptr(), ys[i].data_ptr(), 1, flags,
                                         None, 0, sp)
                assert rc == 0, lib.vptq_last_error()
                if dist is not None:
                    dist.all_reduce(ys[i])      # sum of the ranks' partial outputs
    elif a.mode == "tp":
        launches_per_step = R
        if H // 8 % world:
            raise SystemExit("tp mode needs the vector-row count divisible by the world size")

        def one_pass():
            sp = torch.cuda.current_stream().cuda_stream
            for i in range(R):
                rc = lib.vptq_quant_gemv(descs[i], x.data_ptr(), ys[i].data_ptr(), 1, flags,
                                         None, 0, sp)
                assert rc == 0, lib.vptq_last_error()
                if dist is not None:
                    dist.all_gather_into_tensor(y_full, ys[i].view(-1))
    elif a.mode == "single":
        launches_per_step = R

        def one_pass():
            sp = torch.cuda.current_stream().cuda_stream
            for i in range(R):
                rc = lib.vptq_quant_gemv(descs[i], x.data_ptr(), ys[i].data_ptr(), 1, flags,
                                         None, 0, sp)
                assert rc == 0, lib.vptq_last_error()
    else:
        chunks = []
        for i0 in range(0, R, a.group):
            m = min(a.group, R - i0)
            chunks.append((m, (B.LayerDesc * m)(*descs[i0:i0 + m]),
                           (C.c_void_p * m)(*[x.data_ptr()] * m),
                           (C.c_void_p * m)(*[t.data_ptr() for t in ys[i0:i0 + m]])))
        launches_per_step = len(chunks)
        kname = lib.vptq_quant_gemv_grouped_kernel_name(chunks[0][1], chunks[0][0], 1, flags).decode()

        def one_pass():
            sp = torch.cuda.current_stream().cuda_stream
            for m, arr, xp, yp in chunks:
                rc = lib.vptq_quant_gemv_grouped(arr, m, xp, yp, 1, flags, sp)
                assert rc == 0, lib.vptq_last_error()

    class _Eager:  # same interface as a captured graph
        def replay(self):
            one_pass()

    with torch.cuda.stream(stream):
        one_pass()
        torch.cuda.synchronize()
        captured = True
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                one_pass()
        except Exception:
            if not tp:
                raise
            captured, graph = False, _Eager()  # collectives that refuse capture: eager launches
            torch.cuda.synchronize()
        for _ in range(a.warmup):
            graph.replay()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        for _ in range(a.steps):
            graph.replay()
        e1.record(stream)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    ev_ms = e0.elapsed_time(e1)
    wall, ev_ms = reduce_times(wall, ev_ms, dist, dev)

    ab = alg_bytes(H)
    if tp:   # strong scaling: the ring is processed once per step by all ranks together
        value = job_throughput_gbps(1, ab, R, a.steps, wall)
    else:
        value = job_throughput_gbps(world, ab, R, a.steps, wall)
    us_per_launch = ev_ms * 1e3 / (a.steps * launches_per_step)
    bytes_per_launch = ab * R / launches_per_step / (world if tp else 1)
    achieved = bytes_per_launch / us_per_launch / 1e3     # GB/s
    out = {
        "metric": "decode GEMV effective GB/s (VQuantLinear 2-bit, batch 1)",
        "value": value, "unit": "GB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": wall * 1e3 / a.steps, "higher_is_better": True,
        "scaling": "strong" if tp else "weak",
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"VQuantLinear {H}x{H} v=8 k=256+256 (2-bit) batch=1 seq=1 fp16, "
                               f"ring of {R} distinct layers per GPU ({R * idx_bytes >> 20} MiB of "
                               f"packed indices), 1 step = 1 pass over the ring",
                   "hidden": H, "ring": R, "mode": a.mode, "launches_per_step": launches_per_step,
                   "kernel": kname,
                   "arithmetic": "reference roundings per weight (VPTQ_GEMV_EXACT), fp32 accumulate"
                                 if a.exact else "folded fp32 (default): sum (c+r)*f16(s*x) + sum b*x",
                   "read_ahead_next_layer": bool(a.prefetch),
                   "hipgraph": captured,
                   "parallelism": (f"tp{world}: output rows of every layer split over {world} ranks, "
                                   "RCCL all-gather per layer") if a.mode == "tp" else
                                  (f"tp{world} row-parallel: input columns of every layer split over "
                                   f"{world} ranks, RCCL all-reduce of the partial outputs per layer")
                                  if a.mode == "tp_row" else
                                  f"{world} x independent rings (no collective)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                     "bytes_per_launch": bytes_per_launch, "us_per_launch": us_per_launch,
                     "note": "us_per_launch = HIP-event time over the timed region / launches "
                             "(includes the ~1.8 us inter-kernel gap); the kernels are bound by "
                             "instruction issue / LDS gathers, not by HBM: see compute_floor and "
                             "DESIGN.md §4",
                     "compute_floor": compute_floor(kname, H)},
    }
    # the newest round's summary of this configuration
    rounds = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if d[:1] == "r" and d[1:].isdigit())
    pmc = ""
    for rd in reversed(rounds):
        cand = os.path.join(ROOT, "profiles", rd, f"bench_h{H}_{a.mode}_pmc_summary.json")
        if os.path.exists(cand):
            pmc = cand
            break
    if pmc and not a.exact and not a.prefetch:
        # HBM bytes per launch from a separate rocprofv3 --pmc run of this same command
        # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); see the file's _note
        out["roofline"]["traffic"] = json.load(open(pmc)).get("hbm_bytes_corrected")
        out["roofline"]["traffic_source"] = os.path.relpath(pmc, ROOT)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        torch.cuda.synchronize()
        base, rel = cpu_baseline(layers[0], x, ys[0], H)
        if base is not None:
            out["cpu_baseline"] = base
            out["parity_rel_err_vs_cpu_oracle"] = rel
            assert rel <= 1e-3, f"GPU result differs from the CPU oracle: {rel}"
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

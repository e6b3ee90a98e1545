#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the path-tracing hot path on the BASELINE.json workload.

A "step" is one pass of the hot path over one frame: the z-up RTIOW random-sphere scene
(486 spheres, SURVEY.md §8(d) "S-rtiow"), 1920x1080, 500 spp, 50 bounces — BASELINE.json
configs[2] (the configuration the metric is quoted on).  Scene and CameraData are built once by
the host mirror and are resident on the GPU before the timed region; the timed region is
K x [rt_render (+ for N > 1 one RCCL gather of the row bands to rank 0)].

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL).  The frame's rows are
sharded in interleaved 8-row bands, every rank renders its rows of the SAME frame and one
dist.gather assembles the frame on rank 0 — total work is fixed, so "scaling" is "strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-practice_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import frame_parallel as fp  # noqa: E402
import rtp_bindings as rb  # noqa: E402

WIDTH, HEIGHT, SPP, DEPTH = 1920, 1080, 500, 50
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def host_threads():
    """Host cores this process may use, capped at the 16-core share of a one-GPU box."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def cpu_baseline(host, threads, budget_s=12.0, frame=None, gpu_cam=None):
    """The oracle (CPU restatement of the reference's render_cpu) on this box's host cores, on a
    bounded sample of the same workload: full 1920x1080 frame geometry, depth 50, reduced spp.
    Also returns the traversal statistics that price the algorithmic bytes per sample and, since
    the CPU path is running anyway, renders two rows at the timed configuration's full spp and
    compares them bit for bit with the GPU frame that was just timed."""
    import oracle_bindings as ob   # the checker; never on the product path
    # single thread (what the reference's own CPU path uses): a 1080/40-row slice at 1 spp
    cam1 = rb.rtiow_camera(WIDTH, HEIGHT, 1, DEPTH)
    rows = list(range(0, HEIGHT, 40))
    t0 = time.perf_counter()
    n1 = 0
    for r in rows:
        ob.render(host, cam1, row0=r, row1=r + 1, threads=1)
        n1 += WIDTH
        if time.perf_counter() - t0 > budget_s / 2:
            break
    single = n1 / (time.perf_counter() - t0) / 1e6
    # all cores: whole frame at the spp that fits the budget
    spp = max(1, min(8, int(single * threads * 0.8 * (budget_s / 2) * 1e6 / (WIDTH * HEIGHT))))
    cam = rb.rtiow_camera(WIDTH, HEIGHT, spp, DEPTH)
    t0 = time.perf_counter()
    _, st = ob.render(host, cam, threads=threads, want_stats=True)
    dt = time.perf_counter() - t0
    multi = WIDTH * HEIGHT * spp / dt / 1e6
    checked = None
    if frame is not None and gpu_cam is not None:
        rows = [HEIGHT // 3, HEIGHT - 7]
        checked = {"rows": rows, "gpu_frame_bit_identical": bool(all(
            np.array_equal(ob.render(host, gpu_cam, row0=r, row1=r + 1, threads=threads).view(np.uint32),
                           frame[r:r + 1].view(np.uint32)) for r in rows))}
    return {
        "value": round(multi, 3), "unit": "Msamples/s", "cores": threads, "kind": "port",
        "sample": f"oracle render_cpu restatement, full {WIDTH}x{HEIGHT} frame, depth {DEPTH}, {spp} spp "
                  f"({WIDTH * HEIGHT * spp} samples, {dt:.1f} s, {threads} threads); single thread on "
                  f"{n1} samples of the same frame",
        "single_thread_value": round(single, 4),
        "checked_rows": checked,
    }, st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=SPP, help="override samples per pixel (invalidates the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("RTP_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of the N > 1 flow on one GPU
    if world > 1 and backend != "nccl":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    elif world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device(f"cuda:{local_rank}"))
        except TypeError:       # older torch: no device_id keyword
            dist.init_process_group("nccl", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    local_rank %= torch.cuda.device_count()          # (rehearsals may put several ranks on one GPU)
    torch.cuda.set_device(local_rank)
    rb._check(rb.amd_lib().rt_set_device(local_rank), "rt_set_device")

    host = rb.HostScene.rtiow()
    cam = rb.rtiow_camera(WIDTH, HEIGHT, args.spp, DEPTH)
    dev = rb.DeviceScene(host)                      # scene resident in HBM before timing
    band = fp.DEFAULT_BAND_ROWS
    shard = fp.shard_for_rank(rank, world, band) if world > 1 else None
    local_rows = rb.amd_lib().rt_shard_rows(HEIGHT, C.byref(shard) if shard else None)
    fb = torch.zeros((local_rows, WIDTH, 3), dtype=torch.float32, device=f"cuda:{local_rank}")
    stream = torch.cuda.current_stream().cuda_stream

    kernel_ms, trace_ms, launches, guarded, flagged, rework_ms = [], [], [], [], [], []

    def step(record):
        dev.render(cam, fb.data_ptr(), shard=shard, stream=stream, sync=False)
        if world > 1 and backend != "nccl":
            torch.cuda.synchronize()
            frame = fp.gather_frame(fb.cpu(), HEIGHT, band)          # gloo rehearsal: staged through the host
        else:
            frame = fp.gather_frame(fb, HEIGHT, band) if world > 1 else fb
        if record:
            t = dev.last_timing()     # hipEvent pairs recorded on `stream`: whole call, and around each trace launch
            kernel_ms.append(t.kernel_ms)
            trace_ms.append(t.trace_ms)
            launches.append(t.trace_launches)
            guarded.append(t.guarded)
            flagged.append(t.flagged_samples)
            rework_ms.append(t.rework_ms)
        return frame

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    frame = None
    for _ in range(args.steps):
        frame = step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        samples_per_step = WIDTH * HEIGHT * args.spp
        ms_per_step = elapsed * 1e3 / max(args.steps, 1)
        value = samples_per_step / (ms_per_step * 1e-3) / 1e6
        out = {
            "metric": "Msamples/sec (pixels x spp / s), path-traced frame",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"S-rtiow random-sphere scene (486 spheres, 971 BVH nodes, seed 12345), {WIDTH}x{HEIGHT}, "
                                   f"{args.spp} spp, {DEPTH} bounces, background (0.7,0.8,1.0)",
                       "parallelism": f"row-band shard x{world} + 1 gather" if world > 1 else "single GPU",
                       "traversal": ("guarded near-first walk + exact re-walk of flagged samples" if guarded and guarded[0]
                                     else "reference-order (threaded) walk")},
        }
        base, st = (None, None)
        if world > 1 and os.environ.get("RTP_BENCH_CHECK"):
            import oracle_bindings as ob       # rehearsal only: the assembled frame against the checker
            got = frame.detach().cpu().numpy()
            rows = [HEIGHT // 3, HEIGHT - 7]
            out["checked_rows"] = {"rows": rows, "assembled_frame_bit_identical": bool(all(
                np.array_equal(ob.render(host, cam, row0=r, row1=r + 1, threads=host_threads()).view(np.uint32),
                               got[r:r + 1].view(np.uint32)) for r in rows))}
        if world == 1 and not args.no_cpu_baseline:
            base, st = cpu_baseline(host, host_threads(), frame=frame.detach().cpu().numpy(), gpu_cam=cam)
            out["cpu_baseline"] = base
        # roofline of the dominant (only) kernel: algorithmic bytes per launch / mean launch time
        bytes_per_sample = st.bytes_per_sample(args.spp) if st is not None else 5790.0   # SURVEY.md §8(d) if not re-counted
        local_samples = local_rows * WIDTH * args.spp
        k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
        tr_ms = float(np.mean(trace_ms)) if trace_ms else float("nan")
        n_launch = int(launches[0]) if launches else 0
        # one frame = n_launch launches of the trace kernel (a pass of samples per pixel each);
        # achieved = algorithmic bytes of one launch / its mean duration
        launch_ms = tr_ms / max(n_launch, 1)
        achieved = bytes_per_sample * (local_samples / max(n_launch, 1)) / (launch_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath) and world == 1 and args.spp == SPP:
            try:
                tj = json.load(open(tpath))       # measured for one launch shape: only quoted when this run has it
                traffic = tj.get("bytes_per_trace_launch") if tj.get("trace_launches_per_frame") == n_launch else None
            except Exception:
                traffic = None
        valu = None
        vpath = os.path.join(ROOT, "profiles", "valu_util.json")
        if os.path.exists(vpath) and world == 1 and args.spp == SPP:
            try:
                v = json.load(open(vpath))
                valu = {k: v[k] for k in ("valu_issue_utilisation", "valu_lane_utilisation", "valu_roofline_frac")}
            except Exception:
                valu = None
        out["roofline"] = {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "kernel": "rtk::render_kernel<true,false>" if guarded and guarded[0] else "rtk::render_kernel<true,true>",
            "launches_per_step": n_launch, "launch_ms": round(launch_ms, 3),
            "rework_launch_ms": round(float(np.mean(rework_ms)) / max(n_launch, 1), 3) if rework_ms else None,
            "flagged_sample_fraction": round(float(np.mean(flagged)) / max(local_samples, 1), 6) if flagged else None,
            "step_kernels_ms": round(k_ms, 3),
            "algorithmic_bytes_per_sample": round(bytes_per_sample, 1),
            "valu_side_from_committed_pmc": valu,
            "note": "algorithmic bytes (node/sphere/material records the reference's traversal touches) are served "
                    "from LDS, not HBM; see DESIGN.md 'Roofline' for the VALU-side reading",
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the path-tracing hot path on the BASELINE.json workload.

A "step" is one pass of the hot path over one frame: the z-up RTIOW random-sphere scene
(486 spheres, SURVEY.md §8(d) "S-rtiow"), 1920x1080, 500 spp, 50 bounces — BASELINE.json
configs[2] (the configuration the metric is quoted on).  Scene and CameraData are built once by
the host mirror and are resident on the GPU before the timed region; the timed region is
K x [rt_render (+ for N > 1 one RCCL gather of the row bands to rank 0)].

Process layout
  * `python bench.py --gpus N` with no WORLD_SIZE in the environment is a LAUNCHER: it never touches
    the GPU itself, starts N worker processes (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set,
    rendezvous on 127.0.0.1), relays rank 0's JSON line and exits non-zero if any worker fails.
    At N = 1 it also runs three short `rocprofv3 --pmc` passes of one frame (VALU counters, FETCH_SIZE,
    WRITE_SIZE — separate passes, no tracing) and folds the measured VALU utilisation and HBM traffic
    of the dominant kernel into the line's `roofline` block.
  * with WORLD_SIZE set (the driver's `python -m torch.distributed.run … bench.py --gpus N`, or a
    worker of the launcher above) the process is ONE RANK: torch.distributed over RCCL ("nccl").

N > 1: the frame's rows are sharded in interleaved 8-row bands, every rank renders its rows of the
SAME frame and one dist.gather assembles the frame on rank 0 — total work is fixed, so "scaling"
is "strong".

Prints ONE JSON line (rank 0 / the launcher).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-practice_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

WIDTH, HEIGHT, SPP, DEPTH = 1920, 1080, 500, 50          # the headline workload; set_workload() switches the module to another one
WORKLOAD = "headline"
WORKLOADS = {
    # BASELINE.json configs[2]: the configuration the metric is quoted on
    "headline": dict(width=1920, height=1080, spp=500, half_extent=11, textured_quad=False, texture_size=0,
                     text="S-rtiow random-sphere scene (486 spheres, 971 BVH nodes, seed 12345)"),
    # BASELINE.json configs[4]: the stress scene (a secondary line, `--workload c5`; not what the driver runs)
    "c5": dict(width=3840, height=2160, spp=1000, half_extent=158, textured_quad=True, texture_size=2048,
               text="S-100k stress scene (99857 spheres + one textured METAL quad, 199715 BVH nodes, seed 12345)"),
}


def set_workload(name):
    global WIDTH, HEIGHT, SPP, WORKLOAD
    w = WORKLOADS[name]
    WIDTH, HEIGHT, SPP, WORKLOAD = w["width"], w["height"], w["spp"], name


def make_host_scene():
    import rtp_bindings as rb
    w = WORKLOADS[WORKLOAD]
    if WORKLOAD == "headline":
        return rb.HostScene.rtiow()
    return rb.HostScene.rtiow(half_extent=w["half_extent"], textured_quad=w["textured_quad"], texture_size=w["texture_size"])
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6e12 lane-operations per second
# (= the 157.3 TFLOPS fp32 vector peak at 2 flops per fused lane-op)
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12
NUM_SIMDS = 1024

PMC_VALU = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
            "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"]


# ------------------------------------------------------------------------------------------------
# launcher (parent): no torch.cuda, no rt_* call in this process
# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker_cmd(args, extra=()):
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--spp", str(args.spp), "--workload", args.workload]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.dry_run:
        cmd.append("--dry-run")
    return cmd + list(extra)


def _last_json_line(text):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None


def _pmc_pass(args, counters, tag):
    """One `rocprofv3 --pmc` pass (counters only: never combined with tracing) around ONE frame of the
    same workload; returns {kernel name: {counter: mean per launch}} or None."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    out_dir = tempfile.mkdtemp(prefix=f"rtp_pmc_{tag}_", dir="/tmp")
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", TMPDIR="/tmp", RTP_BENCH_PMC_CHILD="1")
    # the program itself after `--`: the profiler's library has initialised the GPU before it starts
    cmd = [rocprof, "--pmc", *counters, "--output-format", "csv", "-d", out_dir, "-o", "pmc", "--",
           sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "1", "--warmup", "0",
           "--spp", str(args.spp), "--workload", args.workload, "--no-cpu-baseline"]
    try:
        res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=args.pmc_timeout)
    except (subprocess.TimeoutExpired, OSError):
        return None
    if res.returncode != 0:
        sys.stderr.write(f"[bench] rocprofv3 pass '{tag}' failed ({res.returncode}): {res.stderr[-400:]}\n")
        return None
    files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return None
    acc, cnt = {}, {}
    for row in csv.DictReader(open(files[0])):
        key = (row["Kernel_Name"], row["Counter_Name"])
        acc[key] = acc.get(key, 0.0) + float(row["Counter_Value"])
        cnt[key] = cnt.get(key, 0) + 1
    keep = os.environ.get("RTP_BENCH_PMC_KEEP")      # tools/profile_bench.sh: copy the raw CSV for profiles/
    if keep:
        os.makedirs(keep, exist_ok=True)
        shutil.copy(files[0], os.path.join(keep, f"pmc_{tag}.csv"))
    shutil.rmtree(out_dir, ignore_errors=True)
    per = {}
    for (kn, cn), v in acc.items():
        per.setdefault(kn, {})[cn] = v / cnt[(kn, cn)]
        per[kn]["_launches"] = cnt[(kn, cn)]
    return per


def _match_kernel(per, name):
    """Counter rows of the kernel whose (possibly truncated) name matches `name`."""
    if not per:
        return None
    for kn, vals in per.items():
        if kn.startswith(name) or name.startswith(kn) or name in kn:
            return vals
    return None


def _fold_pmc(out, args):
    """N = 1: measure the dominant kernel's VALU utilisation and HBM traffic in THIS run."""
    roof = out.get("roofline")
    if not roof:
        return
    kernel = roof.get("kernel", "render_kernel")
    short = kernel.split("(")[0]
    valu = _match_kernel(_pmc_pass(args, PMC_VALU, "valu"), short)
    fetch = _match_kernel(_pmc_pass(args, ["FETCH_SIZE"], "fetch"), short)
    write = _match_kernel(_pmc_pass(args, ["WRITE_SIZE"], "write"), short)
    if valu and valu.get("GRBM_GUI_ACTIVE") and valu.get("SQ_ACTIVE_INST_VALU"):
        cycles = valu["GRBM_GUI_ACTIVE"] / 8.0                         # summed over the 8 XCDs
        issue = 2.0 * valu["SQ_INSTS_VALU"] / (NUM_SIMDS * cycles)     # a wave64 VALU instruction holds its SIMD-32 for 2 cycles
        lanes = valu["SQ_THREAD_CYCLES_VALU"] / (64.0 * valu["SQ_ACTIVE_INST_VALU"])
        frac = issue * lanes
        roof.update({
            "bound": "valu", "frac": round(frac, 4), "peak": round(VALU_PEAK_TLANEOPS, 1), "unit": "Tlane-op/s",
            "achieved": round(frac * VALU_PEAK_TLANEOPS, 2),
            "valu_issue_utilisation": round(issue, 4), "valu_lane_utilisation": round(lanes, 4),
            "valu_wave_instructions_per_launch": valu["SQ_INSTS_VALU"],
            "valu_wave_instructions_per_sample": round(valu["SQ_INSTS_VALU"] / max(roof.get("samples_per_launch", 1), 1), 2),
            # (sky pixels are nobody's work: the same count over the samples the trace kernel actually took)
            "valu_wave_instructions_per_traced_sample": round(valu["SQ_INSTS_VALU"] / max(roof.get("traced_samples_per_launch") or roof.get("samples_per_launch", 1), 1), 2),
            "wave_time_split": {"issuing": round(valu["SQ_ACTIVE_INST_ANY"] / valu["SQ_WAVE_CYCLES"], 3),
                                "s_waitcnt": round(valu["SQ_WAIT_ANY"] / valu["SQ_WAVE_CYCLES"], 3),
                                "issue_stalled": round(valu["SQ_WAIT_INST_ANY"] / valu["SQ_WAVE_CYCLES"], 3)},
            "source": "rocprofv3 --pmc passes run by this bench invocation (one frame each, counters only)",
            "formula": "frac = issue x lanes; issue = 2 x SQ_INSTS_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE/8); lanes = "
                       "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); peak = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz",
        })
    else:
        roof["source"] = "VALU counters unavailable in this run (rocprofv3 pass failed or missing); see profiles/"
    if fetch and write and "FETCH_SIZE" in fetch and "WRITE_SIZE" in write:
        # MI355X_MICROARCH.md §HBM: counters in KiB; gfx950 reports half of a wide coalesced read
        traffic = (2.0 * fetch["FETCH_SIZE"] + write["WRITE_SIZE"]) * 1024.0
        roof["traffic"] = int(traffic)
        launch_s = roof["launch_ms"] * 1e-3
        roof["hbm_measured"] = {"bytes_per_launch": int(traffic), "GBps": round(traffic / launch_s / 1e9, 1),
                                "frac_of_peak": round(traffic / launch_s / 1e9 / HBM_PEAK_GBS, 4),
                                "fetch_kib": fetch["FETCH_SIZE"], "write_kib": write["WRITE_SIZE"]}


def launcher(args):
    n = args.gpus
    port = _free_port()
    procs = []
    for rank in range(n):
        env = dict(os.environ, WORLD_SIZE=str(n), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(_worker_cmd(args), env=env, stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL,
                                      stderr=None, text=True))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    if any(codes):
        sys.stderr.write(f"[bench] worker exit codes {codes}\n")
        sys.stdout.write(out0 or "")
        return 1
    out = _last_json_line(out0 or "")
    if out is None:
        sys.stderr.write("[bench] rank 0 printed no JSON line\n")
        sys.stdout.write(out0 or "")
        return 1
    if out.get("n_gpus") != n:
        sys.stderr.write(f"[bench] rank 0 reports n_gpus={out.get('n_gpus')} but --gpus {n}\n")
        return 1
    if n == 1 and not args.no_pmc and not args.dry_run:
        _fold_pmc(out, args)
    print(json.dumps(out), flush=True)
    return 0


# ------------------------------------------------------------------------------------------------
# worker: one rank
# ------------------------------------------------------------------------------------------------
def cpu_model():
    """Model string of the host CPU (the CPU baseline states core count AND model)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_threads():
    """Host cores this process may use, capped at the 16-core share of a one-GPU box."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def cpu_baseline(host, threads, budget_s=12.0, frame=None, gpu_cam=None, dev=None):
    """The oracle (CPU restatement of the reference's render_cpu) on this box's host cores, on a
    bounded sample of the same workload: full 1920x1080 frame geometry, depth 50, reduced spp.
    Also returns the traversal statistics that price the algorithmic bytes per sample and, since
    the CPU path is running anyway, renders two rows at the timed configuration's full spp and
    compares them bit for bit with the GPU frame that was just timed."""
    import numpy as np
    import oracle_bindings as ob   # the checker; never on the product path
    import rtp_bindings as rb
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
        check_spp = gpu_cam.samples_per_pixel
        if WORKLOAD != "headline" and dev is not None:
            # the stress scene: a row at 1000 spp is 20 s of one host thread — the same scene handle renders the frame once
            # more at 8 spp and two rows of THAT are compared
            check_spp = 8
            gpu_cam = rb.rtiow_camera(WIDTH, HEIGHT, check_spp, DEPTH)
            frame, _ = dev.render_to_host(gpu_cam)
        checked = {"rows": rows, "spp": check_spp, "gpu_frame_bit_identical": bool(all(
            np.array_equal(ob.render(host, gpu_cam, row0=r, row1=r + 1, threads=threads).view(np.uint32),
                           frame[r:r + 1].view(np.uint32)) for r in rows))}
    return {
        "value": round(multi, 3), "unit": "Msamples/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
        "sample": f"oracle render_cpu restatement, full {WIDTH}x{HEIGHT} frame, depth {DEPTH}, {spp} spp "
                  f"({WIDTH * HEIGHT * spp} samples, {dt:.1f} s, {threads} threads); single thread on "
                  f"{n1} samples of the same frame",
        "single_thread_value": round(single, 4),
        "checked_rows": checked,
    }, st


def dry_run_worker(args, world, rank):
    """No GPU: the N-rank control flow only (rendezvous, band arithmetic, one gather, world-size check).
    Every rank fills its rows with a function of (row, column) and rank 0 checks the assembled frame.
    The line it prints carries no measurement."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import frame_parallel as fp
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    band = fp.DEFAULT_BAND_ROWS
    rows = fp.shard_row_indices(HEIGHT, band, world, rank)
    pattern = lambda r: (r[:, None, None] * 4096.0 + np.arange(WIDTH)[None, :, None] + np.arange(3)[None, None, :] * 0.25).astype(np.float32)
    local = torch.from_numpy(pattern(rows))
    frame = fp.gather_frame(local, HEIGHT, band) if world > 1 else local
    if rank == 0:
        ok = bool(np.array_equal(frame.numpy(), pattern(np.arange(HEIGHT))))
        print(json.dumps({"metric": "Msamples/sec (pixels x spp / s), path-traced frame", "value": None, "unit": "Msamples/s",
                          "n_gpus": world, "steps": 0, "warmup": 0, "dry_run": True, "backend": "gloo",
                          "world_size_seen_by_collective": dist.get_world_size() if world > 1 else 1,
                          "assembled_frame_ok": ok, "scaling": "strong"}), flush=True)
        if not ok:
            raise SystemExit(2)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def worker(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run_worker(args, world, rank)

    import ctypes as C
    import numpy as np
    import torch
    import torch.distributed as dist
    import frame_parallel as fp
    import rtp_bindings as rb

    backend = os.environ.get("RTP_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of the N > 1 flow on one GPU
    if world > 1 and backend != "nccl":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    elif world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device(f"cuda:{local_rank % max(torch.cuda.device_count(), 1)}"))
        except TypeError:       # older torch: no device_id keyword
            dist.init_process_group("nccl", rank=rank, world_size=world)
    if world > 1:
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    local_rank %= torch.cuda.device_count()          # (rehearsals may put several ranks on one GPU)
    torch.cuda.set_device(local_rank)
    rb._check(rb.amd_lib().rt_set_device(local_rank), "rt_set_device")

    host = make_host_scene()
    cam = rb.rtiow_camera(WIDTH, HEIGHT, args.spp, DEPTH)
    # scene resident in HBM before timing.  reuse_view_lists = -1: every timed frame makes its per-pixel candidate lists itself —
    # the library's default keeps them for a repeated view, and a benchmark step is ALL of a frame's work
    dev = rb.DeviceScene(host, reuse_view_lists=-1)
    band = fp.DEFAULT_BAND_ROWS
    shard = fp.shard_for_rank(rank, world, band) if world > 1 else None
    local_rows = rb.amd_lib().rt_shard_rows(HEIGHT, C.byref(shard) if shard else None)
    fb = torch.zeros((local_rows, WIDTH, 3), dtype=torch.float32, device=f"cuda:{local_rank}")
    stream = torch.cuda.current_stream().cuda_stream
    # every buffer of the collective exists before the timed region (frame_parallel.FrameGatherer)
    gatherer = fp.FrameGatherer(HEIGHT, WIDTH, band, device=("cpu" if backend != "nccl" else f"cuda:{local_rank}")) if world > 1 else None

    kernel_ms, trace_ms, launches, guarded, flagged, rework_ms, primary_ms, gather_ms = [], [], [], [], [], [], [], []
    last_t = [None]

    def step(record):
        dev.render(cam, fb.data_ptr(), shard=shard, stream=stream, sync=False)
        g0 = 0.0
        if world > 1 and record:
            torch.cuda.synchronize()           # (recorded steps only: separates this rank's render from the collective)
            g0 = time.perf_counter()
        if world > 1 and backend != "nccl":
            torch.cuda.synchronize()
            frame = gatherer.gather(fb.cpu())                        # gloo rehearsal: staged through the host
        else:
            frame = gatherer.gather(fb) if world > 1 else fb
        if record:
            if world > 1:
                torch.cuda.synchronize()
                gather_ms.append((time.perf_counter() - g0) * 1e3)
            t = dev.last_timing()     # hipEvent pairs recorded on `stream`: whole call, and around each trace launch
            last_t[0] = t
            kernel_ms.append(t.kernel_ms)
            trace_ms.append(t.trace_ms)
            launches.append(t.trace_launches)
            guarded.append(t.guarded)
            flagged.append(t.flagged_samples)
            rework_ms.append(t.rework_ms)
            primary_ms.append(t.primary_ms)
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

    # ---- second clock: what the reference's own timer brackets besides the kernel — the copy of the float
    # frame to the host (src/camera.cu:333-343 times render_kernel + cudaMemcpy D2H + the saver loop); never `value`
    ref_equiv_ms = None
    host_frame = torch.empty((HEIGHT, WIDTH, 3), dtype=torch.float32).pin_memory() if rank == 0 else None
    fence()
    t1 = time.perf_counter()
    reps = min(args.steps, 3)
    for _ in range(reps):
        f = step(False)
        if rank == 0:
            host_frame.copy_(f)          # blocking device → host copy of the assembled float frame
    fence()
    if rank == 0:
        ref_equiv_ms = (time.perf_counter() - t1) * 1e3 / max(reps, 1)

    if rank == 0:
        samples_per_step = WIDTH * HEIGHT * args.spp
        ms_per_step = elapsed * 1e3 / max(args.steps, 1)
        value = samples_per_step / (ms_per_step * 1e-3) / 1e6
        out = {
            "metric": "Msamples/sec (pixels x spp / s), path-traced frame",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "reference_equivalent_ms": round(ref_equiv_ms, 3) if ref_equiv_ms else None,
            "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{WORKLOADS[WORKLOAD]['text']}, {WIDTH}x{HEIGHT}, "
                                   f"{args.spp} spp, {DEPTH} bounces, background (0.7,0.8,1.0)",
                       "parallelism": f"row-band shard x{world} + 1 gather" if world > 1 else "single GPU",
                       "collective_backend": (backend if world > 1 else None),
                       "world_size_seen_by_collective": (dist.get_world_size() if world > 1 else 1),
                       "traversal": ("guarded near-first walk + exact re-walk of flagged samples" if guarded and guarded[0]
                                     else "reference-order (threaded) walk"),
                       "primary_visibility": bool(last_t[0] is not None and last_t[0].primary_visibility),
                       "front_primitives": int(last_t[0].front_primitives) if last_t[0] is not None else 0,
                       "wide_nodes": bool(last_t[0] is not None and last_t[0].wide_nodes),
                       "view_lists_reused_between_steps": False},
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
            base, st = cpu_baseline(host, host_threads(), frame=frame.detach().cpu().numpy(), gpu_cam=cam, dev=dev)
            out["cpu_baseline"] = base
        # ---- roofline of the dominant kernel (the trace launch) ------------------------------------
        # The path is VALU-bound (scene tables live in LDS); the launcher adds the measured VALU issue x lane
        # utilisation (`bound`, `frac`, `achieved`, `peak`) and the PMC HBM bytes (`traffic`) to this block.
        bytes_per_sample = st.bytes_per_sample(args.spp) if st is not None else 5790.0   # SURVEY.md §8(d) if not re-counted
        local_samples = local_rows * WIDTH * args.spp
        k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
        tr_ms = float(np.mean(trace_ms)) if trace_ms else float("nan")
        n_launch = int(launches[0]) if launches else 0
        launch_ms = tr_ms / max(n_launch, 1)
        alg = bytes_per_sample * (local_samples / max(n_launch, 1)) / (launch_ms * 1e-3) / 1e9
        committed = None
        vpath = os.path.join(ROOT, "profiles", "valu_util.json")
        if os.path.exists(vpath):
            try:
                v = json.load(open(vpath))
                committed = {k: v[k] for k in ("valu_issue_utilisation", "valu_lane_utilisation", "valu_roofline_frac")}
            except Exception:
                committed = None
        out["roofline"] = {
            "bound": "valu", "achieved": None, "peak": round(VALU_PEAK_TLANEOPS, 1), "unit": "Tlane-op/s", "frac": None,
            "traffic": None,
            "kernel": dev.trace_kernel_name(),
            "launches_per_step": n_launch, "launch_ms": round(launch_ms, 3),
            "kernel_resources": ({"vgprs_per_lane": int(last_t[0].trace_vgprs), "scratch_bytes_per_lane": int(last_t[0].trace_scratch_bytes),
                                  "workgroup_size": int(last_t[0].workgroup_size), "workgroups": int(last_t[0].num_workgroups),
                                  "lds_bytes_per_workgroup": int(last_t[0].lds_bytes),
                                  "waves_per_simd": int(last_t[0].workgroup_size) // 64 * int(last_t[0].num_workgroups)
                                                    // max(torch.cuda.get_device_properties(local_rank).multi_processor_count, 1) // 4,
                                  "source": "hipFuncGetAttributes of the loaded code object + the launch shape of this run"}
                                 if last_t[0] is not None else None),
            "primary_visibility_pass_ms": round(float(np.mean(primary_ms)), 3) if primary_ms else None,
            "samples_per_launch": local_samples // max(n_launch, 1),
            "traced_samples_per_launch": (int(last_t[0].traced_samples) // max(n_launch, 1)) if last_t[0] is not None else None,
            "abandoned_passes": int(last_t[0].abandoned_passes) if last_t[0] is not None else None,
            "rework_launch_ms": round(float(np.mean(rework_ms)) / max(n_launch, 1), 3) if rework_ms else None,
            "flagged_sample_fraction": round(float(np.mean(flagged)) / max(local_samples, 1), 6) if flagged else None,
            "step_kernels_ms": round(k_ms, 3),
            "per_rank": ({"trace_ms": round(tr_ms, 3), "render_ms": round(k_ms, 3), "gather_ms": round(float(np.mean(gather_ms)), 3),
                          "note": "rank 0's own render (hipEvents) and the collective as rank 0 sees it (wall clock from its render's end: "
                                  "includes waiting for the slowest rank)"} if world > 1 else None),
            "hbm_algorithmic_equiv": {
                "GBps": round(alg, 1), "over_hbm_peak": round(alg / HBM_PEAK_GBS, 3),
                "bytes_per_sample": round(bytes_per_sample, 1),
                "note": "SURVEY.md §8(d) figure: bytes of node/sphere/material records the REFERENCE's walk touches per sample x "
                        "samples per launch / launch time.  These records are served from LDS and the guarded walk touches "
                        "fewer of them, so this is NOT comparable with the HBM peak; kept as a secondary figure only."},
            "committed_profile_valu": committed,
            "source": "no PMC pass in this process (run `python bench.py` without WORLD_SIZE at N=1 for live counters)",
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (invalidates the headline)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="headline",
                    help="headline = BASELINE configs[2] (the metric's configuration, what the driver runs); c5 = BASELINE configs[4], a secondary line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="launcher, N = 1: skip the rocprofv3 counter passes")
    ap.add_argument("--pmc-timeout", type=int, default=240)
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: rehearse the N-rank launch + gather over gloo; prints a line without a measurement")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    set_workload(args.workload)
    if args.spp <= 0:
        args.spp = SPP
    if "WORLD_SIZE" in os.environ:
        worker(args)
        return 0
    return launcher(args)


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (paths x bounces) of the path tracer at 1080p.

    python bench.py --gpus N --steps K --warmup W

Workload (the configuration the headline metric is quoted on): the reference's bundled `cubes` scene (424 triangles, 24 of
them emissive), 1920x1080, RECURSION 6, CLOCK 95834783, default camera, constant environment
(the reference's .exr environment is not in its repository; data is synthetic in that sense).
The job is ONE image of steps x `--spp-per-step` samples per pixel (4 x 4096 = 16 384 spp by default), whatever the
number of GPUs ("scaling": "strong", the form of the metric "seconds to N spp"): a step is the K-th part of the image's
samples, rendered in passes of 512 samples (the largest pass the library takes: 132 GB of path queues at 1080p; passes
of 256 samples, 66 GB, are 2.5 % slower), so that on 8 GPUs a rank's step is still one full pass.
`seconds_per_1024spp` is the timed region, reduce included, scaled to the 1024 spp of the metric.  Rays =
findIntersection executions (one per bounce entered); light probes are counted separately.

With N > 1 there is one rank per GPU: `python bench.py --gpus N` starts them itself (python -m torch.distributed.run,
as a child process, before anything touches a GPU) unless it already runs under a launcher (RANK in the environment).
The scene is replicated and rank r renders SAMPLE = r, r + N, r + 2N, ...; the per-GPU radiance sums are gathered on
rank 0 over RCCL and added there in rank order, inside the timed region.  Modes:
  default          the steps x spp-per-step samples of one image are split over the ranks ("scaling": "strong")
  --total-spp T    the same with T samples in all (the step count follows from T); with --width 3840 --height 2160
                   --total-spp 4096 this is the 4K / 4096 spp configuration
  --weak           every rank renders steps x spp-per-step samples ("scaling": "weak")

Other workloads (not the headline; `--workload`): `obiwan` = the bundled 12 030-triangle model at RECURSION 16,
`c3` = a synthetic 1 M-triangle height field (gpuraytracer_amd/synthetic.py) ingested through the OBJ/MTL reader,
with 4096^2 albedo / normal / ARM layers that go through OpenEXR files like the reference's loadMaterial.

Rank 0 prints one JSON line.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1920, 1080
SPP_PER_STEP = 4096         # samples of the image per step by default
SPP_PER_PASS = 512          # samples per pass by default; the counter files under profiles/ are measured with it
# Vector-instruction issue of gfx950: a SIMD issues a wave64 instruction over 2 clocks (32 lanes per clock; the
# architectural rate).  Measured by tools/microbench_valu.hip (profiles/r03_valu_ceiling.txt), 8 waves per SIMD, whole
# chip: one per 2.32 clocks for v_fma/v_mul/v_add/v_mov/integer add/logic (and for even mixes of those with the slower
# kinds), one per 4.2 clocks for v_min/v_max/v_cmp/v_cndmask/shifts/conversions/packed f32 on their own.
VALU_ARCH_CLOCKS = 2.0
VALU_FAST_CLOCKS = 2.32
VALU_SLOW_CLOCKS = 4.2
SIMDS, CLOCK_HZ = 1024, 2.4e9
# one operation per lane and vector instruction (the intersection arithmetic may not fuse multiply-adds): 32 lanes per
# clock and SIMD
PEAK_LANE_TOPS = SIMDS * 32 * CLOCK_HZ / 1e12
PEAK_VALU_TFLOPS = PEAK_LANE_TOPS
L2_PEAK_GBPS = 34500.0          # all 8 XCDs (MI355X_MICROARCH.md)
# vector instructions of a traversal step (assembly listing of the traversal loops): a node step of the hierarchy resident in
# LDS, of the hierarchy read from memory (12 byte-selects more: its planes are 15-bit grid positions), a triangle test
# (17 to the plane test, 24 more to the inside test)
INSTS_NODE_RESIDENT, INSTS_NODE_MEMORY, INSTS_TRIANGLE = 32, 46, 41
FLOPS_PER_CLOSEST_TEST = 35     # raytracer.glsl:135-158, full pass
FLOPS_PER_PROBE_TEST = 34       # raytracer.glsl:171-195


def constant_env():
    env = np.zeros((4, 8, 4), dtype=np.float32)
    env[..., 0], env[..., 1], env[..., 2], env[..., 3] = 0.5, 0.7, 1.0, 1.0
    return env


def kernel_source_fingerprint():
    """SHA-256 over the kernel sources: counter files under profiles/ carry the fingerprint of the sources they were
    measured on, and are used only while it still matches."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "gpuraytracer_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".h", ".hip")):
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def cpu_baseline(scene, cam, clock, recursion, budget_s=12.0, size=None, layers=None, atlas=0):
    """The CPU restatement (oracle/, C-library math, OpenMP) on a bounded sample of the same workload:
    whole frames of consecutive SAMPLE indices, as many as fit the time budget.  `size`: a smaller target for the same
    camera (the restatement loops over all triangles per query as the reference does: on a million triangles a 1080p
    frame takes minutes); `layers`: the texture layers of the c3 workload."""
    from oracle import oracle
    w, h = size if size else (W, H)
    sc = oracle.OracleScene(*scene, libm=True)
    sc.upload_env(constant_env())
    if layers is not None:
        sc.set_atlas_size(atlas)
        flags = np.zeros(len(scene[3]), dtype=np.int32)
        for m in (0, 1):
            for k in range(3):
                sc.upload_atlas_layer(m, k, layers[k])
            flags[m] = 1
        sc.set_has_texture(flags)
    t0 = time.perf_counter()
    _, st = sc.render(cam, clock, recursion, 1.0, w, h, 0, 1)
    t1 = time.perf_counter() - t0
    spp = int(max(1, min(32, budget_s / max(t1, 1e-3))))
    t0 = time.perf_counter()
    _, st = sc.render(cam, clock, recursion, 1.0, w, h, 0, spp)
    dt = time.perf_counter() - t0
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    return {"value": st["rays"] / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%dx%d%s, SAMPLE 0..%d (%d rays) in %.2f s, OpenMP over pixels"
                      % (w, h, "" if not size else " (the camera of the %dx%d workload on a smaller target)" % (W, H), spp - 1, st["rays"], dt)}


def llvmpipe_baseline():
    """The reference's own shader under Mesa llvmpipe cannot run on the GPU box (no reference, no Mesa there); the time
    it took in the build container for configuration 1 is carried along with the fixture it produced."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "ref_c1_cubes512.json")) as f:
            c1 = json.load(f)
        secs = float(c1["llvmpipe_seconds_16spp"])
    except (OSError, ValueError, KeyError):
        return None
    rays = float(c1.get("rays_16spp", 0)) or 512 * 512 * 16 * 1.2557
    return {"value": rays / secs / 1e6, "unit": "Mrays/s", "kind": "reference",
            "cores": int(c1.get("llvmpipe_threads", 8)),
            "sample": "cubes 512x512, SAMPLE 0..15 in %.2f s: the reference's raytracer.glsl under Mesa llvmpipe, measured in "
                      "the build container (%d threads), not on this machine" % (secs, int(c1.get("llvmpipe_threads", 8)))}


def load_workload(g, args):
    """Returns (scene tuple, camera, recursion, description, texture layers or None)."""
    if args.workload == "cubes":
        scene = g.read_rtscene(os.path.join(ROOT, "tests", "golden", "scenes", "cubes.rtscene"))
        return scene, g.camera_matrix(), 6, "cubes 424 tris", None
    if args.workload == "obiwan":
        scene = g.read_rtscene(os.path.join(ROOT, "tests", "golden", "scenes", "_ObiWan.rtscene"))
        return scene, g.camera_matrix(), 16, "_ObiWan 12030 tris", None
    import tempfile
    from gpuraytracer_amd import host, synthetic
    with tempfile.TemporaryDirectory() as tmp:
        base = os.path.join(tmp, "grid")
        synthetic.write_displaced_grid(base, args.grid)
        scene = host.load_wavefront(base)
        # the three texture layers travel as OpenEXR files, as the reference's loadMaterial reads them
        # (Scene.cpp:75-102, :206-229: <name>_albedo.exr, _normal.exr, _arm.exr into 4096^2 atlas layers)
        layers = []
        for k, (suffix, pixels) in enumerate(zip(("albedo", "normal", "arm"), synthetic.texture_layers(args.atlas))):
            path = os.path.join(tmp, "synthetic_%s.exr" % suffix)
            host.write_exr(path, pixels[..., :3])
            layers.append(host.read_exr(path))
            os.remove(path)
    cam = g.camera_matrix((0.0, 2.2, -5.5), (-0.45, 0.0))
    return (scene, cam, 6, "synthetic height field %d tris, %d^2 atlas layers from OpenEXR files on 2 materials"
            % (len(scene[0]), args.atlas), layers)


PROFILE_PREFIX = "r03"


def profile_counters(workload, kernel_key):
    """Per-launch counters of one kernel from rocprofv3 passes over this command (separate --pmc runs, made by
    tools/prof_bench.sh and tools/prof_summary.py and kept under profiles/): FETCH_SIZE_KB (to be doubled: gfx950
    reports half of wide reads), WRITE_SIZE_KB, vector instructions.  Returns (counters, provenance) or (None, why)."""
    name = "%s_pmc_counters_%s.json" % (PROFILE_PREFIX, workload)
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None, "no counter file profiles/%s" % name
    meta = table.get("_meta", {})
    if meta.get("kernel_source_fingerprint") != kernel_source_fingerprint():
        return None, "profiles/%s was measured on other kernel sources (fingerprint %s)" % (name, meta.get("kernel_source_fingerprint"))
    for kernel, counters in table.items():
        # the depth-0 kernel has a second instantiation for the later rounds of a pass ("..., true>"), which usually return at once
        if kernel.startswith(kernel_key) and not kernel.endswith(", true>"):
            return counters, "profiles/%s: separate rocprofv3 --pmc passes of `%s`, same kernel sources" % (name, meta.get("command", "?"))
    return None, "profiles/%s has no kernel %s" % (name, kernel_key)


def launch_command(n_gpus, argv, port, script=None):
    """The command that starts one rank per GPU for `python bench.py --gpus N <argv>`."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)


def launch_ranks(n_gpus, argv, script=None):
    """Starts the ranks as a child process (this process has not touched a GPU and does not, it only waits), hands rank 0's
    JSON line on to standard output and returns the child's exit status."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver shares device memory between processes by dmabuf only
    env.setdefault("OMP_NUM_THREADS", "1")
    child = subprocess.run(launch_command(n_gpus, argv, port, script), env=env, stdout=subprocess.PIPE)
    lines = [l for l in child.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    elif child.returncode == 0:
        sys.stderr.write("bench.py: the ranks ended without a result line\n")
        return 1
    return child.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=SPP_PER_STEP,
                    help="samples per pixel of the image that one step renders, all ranks together (with --weak: each rank)")
    ap.add_argument("--total-spp", type=int, default=0,
                    help="samples per pixel of the image instead of steps x spp-per-step (the step count follows from it)")
    ap.add_argument("--weak", action="store_true", help="every rank renders steps x spp-per-step samples (weak scaling)")
    ap.add_argument("--batch-samples", type=int, default=SPP_PER_PASS, help="samples per pass (about 124 bytes of queue memory per pixel-sample of a pass)")
    ap.add_argument("--queue-rounds", type=int, default=-1, help="-1 library default; the path queues hold 1/N of a pass's paths")
    ap.add_argument("--pipeline", type=int, default=1)
    ap.add_argument("--accel", type=int, default=-1, help="-1 auto, 0 loops over all triangles, 1 hierarchy")
    ap.add_argument("--workload", choices=("cubes", "obiwan", "c3"), default="cubes")
    ap.add_argument("--grid", type=int, default=724, help="c3: cells per side of the height field")
    ap.add_argument("--atlas", type=int, default=4096, help="c3: edge of the texture layers")
    ap.add_argument("--refill", type=int, default=-1, help="-1 library default, 0/1 closest-hit kernels without/with lane refill")
    ap.add_argument("--resident", type=int, default=-1, help="-1 library default, 0 never stage a whole hierarchy into LDS")
    ap.add_argument("--share", type=int, default=-1, help="-1 library default; 0/1 any-hit searches shared among the lanes of a wave")
    ap.add_argument("--staged-depths", type=int, default=-2, help="-2 library default; bounces after the first that run staged before the tail kernel (-1 all)")
    ap.add_argument("--width", type=int, default=1920, help="target width (the headline metric is quoted at 1920 x 1080)")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--unordered-reduce", action="store_true", help="N > 1: one RCCL reduce(SUM) instead of gather + add in rank order")
    ap.add_argument("--lane-utilisation", type=int, default=-1,
                    help="0: leave out the traversal work per query and the lane utilisation of the traversal loops (two extra "
                         "32 spp renders with the measurement builds of the library, in child processes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bruteforce-pass", action="store_true")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    global W, H
    W, H = args.width, args.height

    # Libraries underneath write to the process's standard output (RCCL prints a version banner at its first
    # collective); the contract is ONE JSON line there, so everything else goes to standard error.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import gpuraytracer_amd as g
    from gpuraytracer_amd.distributed import reduce_sums, step_plan

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    # GPURT_BENCH_SHARED_GPU=1: a rehearsal of the multi-rank path where there are fewer GPUs than ranks (the ranks share
    # the devices there are and talk over gloo; RCCL refuses two ranks on one device).  It exercises the launcher, the sample
    # split, the ordered reduce and the rank-0 line on hardware; its rate says nothing about scaling.
    shared_gpu = os.environ.get("GPURT_BENCH_SHARED_GPU") == "1"
    device = local_rank % max(1, torch.cuda.device_count()) if shared_gpu else local_rank
    torch.cuda.set_device(device)
    dist = None
    if world > 1 or "RANK" in os.environ:     # under torch.distributed.run also with one rank (exercises the RCCL path)
        import torch.distributed as dist
        if shared_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))

    scene, cam, RECURSION, workload_name, layers = load_workload(g, args)
    cpu_layers = None
    clock = g.DEFAULT_CLOCK
    strong = not args.weak
    if strong:
        # one image of total_spp samples; this rank's share of it, in steps of `spp` samples per rank (the K-th part of
        # the share of rank 0); every rank takes the same number of steps
        total_spp = args.total_spp if args.total_spp > 0 else args.steps * args.spp_per_step
        n_steps = max(1, -(-args.total_spp // args.spp_per_step)) if args.total_spp > 0 else max(1, args.steps)
        spp = max(1, -(-(-(-total_spp // world)) // n_steps))
        plan = step_plan(total_spp, rank, world, spp)
    else:
        spp = args.spp_per_step
        total_spp = args.steps * spp * world
        plan = [(rank + world * (i * spp), spp, world) for i in range(args.steps)]
    steps = len(plan)
    my_count = sum(n for _, n, _ in plan)

    ctx = g.Context(device)
    ctx.set_option("pipeline", args.pipeline)
    ctx.set_option("batch_samples", args.batch_samples)
    ctx.set_option("accel", args.accel)
    if args.queue_rounds > 0:
        ctx.set_option("queue_rounds", args.queue_rounds)
    if args.refill >= 0:
        ctx.set_option("refill", args.refill)
    if args.resident >= 0:
        ctx.set_option("resident", args.resident)
    if args.share >= 0:
        ctx.set_option("share", args.share)
    if args.staged_depths >= -1:
        ctx.set_option("staged_depths", args.staged_depths)
    ctx.set_option("profiling", 1)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    t_up = time.perf_counter()
    ctx.upload_scene(*scene)
    upload_s = time.perf_counter() - t_up      # includes building the hierarchy on the host
    if layers is not None:
        ctx.set_atlas_size(args.atlas)
        flags = np.zeros(len(scene[3]), dtype=np.int32)
        for m in (0, 1):
            for k in range(3):
                ctx.upload_atlas_layer(m, k, layers[k])
            flags[m] = 1
        ctx.set_has_texture(flags)
        cpu_layers = layers if not args.no_cpu_baseline else None       # (the CPU baseline of this workload samples them too)
        layers = None
    ctx.upload_env(constant_env())
    ctx.resize(W, H)
    sums = torch.zeros(H * W * 4, dtype=torch.float32, device="cuda")     # accumulation buffer (float4 per pixel)
    ctx.set_sum_buffer(sums.data_ptr())
    ctx.set_frame(cam, clock, RECURSION, 1.0)

    def step(i):
        # SAMPLE indices of this rank: rank + world * k; warm-up steps (i >= steps) render samples beyond the timed ones
        begin, n, stride = plan[i] if i < steps else (rank + world * (i * spp), spp, world)
        if n > 0:
            ctx.trace(begin, n, stride)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(steps + i)
    if dist is not None:                       # the first collective sets up the communicator: not inside the timed region
        reduce_sums(sums, dst=0, ordered=not args.unordered_reduce)
    barrier()
    sums.zero_()
    ctx.reset_stats()
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    t_render = time.perf_counter()
    if dist is not None:
        reduce_sums(sums, dst=0, ordered=not args.unordered_reduce)
    barrier()
    t_end = time.perf_counter()
    dt = t_end - t0
    reduce_ms = (t_end - t_render) * 1e3       # reduce and the closing barrier (which also absorbs the ranks' skew)

    st = ctx.stats()
    tot = torch.tensor([float(st["rays"]), float(st["shadow_probes"] + st["glossy_probes"]), dt, reduce_ms, float(my_count)],
                       dtype=torch.float64, device="cpu" if shared_gpu else "cuda")
    if dist is not None:
        tmax = tot.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt, reduce_ms = float(tmax[2]), float(tmax[3])
    rays, probes, total_samples = float(tot[0]), float(tot[1]), int(tot[4])

    if rank == 0:
        img = sums.view(H, W, 4)[..., :3] / float(total_samples)
        finite = torch.isfinite(img).all(dim=2)
        # the shader's sqrt(1 - cos^2) (raytracer.glsl:204) goes NaN when a rounded cosine exceeds 1; such
        # samples poison their pixel in the reference too, and are reproduced here
        nonfinite_pixels = int((~finite).sum().item())
        mean = img[finite].mean().item()
        # dominant kernel: the depth-0 launches (camera ray, closest hit, shading, probe in place, rest of the bounce);
        # their time comes from HIP events on the context's stream (rt_stats), recorded inside the timed region
        ext_ms, ext_n = st["extend0_ms"], max(1, st["extend0_launches"])
        # the library's own rule (gpurt.h, option "accel"): hierarchy from 64 triangles on unless it is turned off
        used_bvh = args.accel != 0 and len(scene[0]) >= 64
        if used_bvh:
            # Nodes and triangles come from LDS/L1/L2; what has to cross HBM is what the stage leaves behind: a 12-byte
            # radiance record per path that ended, a 64-byte path record per path that goes on (depth1_paths counts those,
            # the ones among the deferred probes included: an upper bound by that much), a 112-byte probe record per
            # probe left to the probe kernels.
            d1, dp, pk = float(st["depth1_paths"]), float(st["deferred_probes"]), float(st["parked_probes"])
            algo = (float(st["pixel_samples"]) - d1 - dp) * 12.0 + d1 * 64.0 + pk * 112.0 + (pk - dp) * 112.0
            achieved = algo / (ext_ms * 1e-3) / 1e9 if ext_ms > 0 else 0.0
            prof, provenance = profile_counters(args.workload, "k_first<")
            comparable = prof is not None and min(args.batch_samples, spp) == SPP_PER_PASS and (W, H) == (1920, 1080)
            launch_s = ext_ms / ext_n * 1e-3 if ext_ms > 0 else 0.0
            traffic = (2.0 * prof["FETCH_SIZE_KB"] + prof["WRITE_SIZE_KB"]) * 1024.0 if comparable else None
            kernel_name = ("k_first (depth 0: camera ray, closest hit by hierarchy, shading, light probe in place with parked "
                           "probes taken up by idle lanes, rest of the bounce)")
            memory_path = {"algorithmic_bytes_per_launch": algo / ext_n, "achieved_GBps": achieved, "hbm_peak_GBps": 8000.0,
                           "hbm_frac": achieved / 8000.0,
                           "note": "12 B per path that ended, 64 B per path that goes on, 112 B written per parked probe and "
                                   "read back per probe taken up again"}
            if comparable and launch_s > 0 and "TCC_HIT_sum" in prof:
                # what the same counter passes say about the memory path during a launch: bytes that crossed the fabric side
                # of the L2s (HBM or Infinity Cache) per second, the L2 requests and hit rate, and how busy the CUs' vector
                # memory front ends (texture addressers) were; hierarchies that do not fit LDS/L2 are bound there
                requests = prof["TCC_HIT_sum"] + prof["TCC_MISS_sum"]
                memory_path.update({
                    "beyond_l2_GBps": traffic / launch_s / 1e9,
                    "beyond_l2_frac_of_hbm_peak": traffic / launch_s / 1e9 / 8000.0,
                    "l2_requests_per_launch": requests,
                    "l2_hit_rate": prof["TCC_HIT_sum"] / max(1.0, requests),
                    # a request moves 64 or 128 bytes (FETCH_SIZE tallies them at 64): the lower figure
                    "l2_GBps": requests * 64.0 / launch_s / 1e9,
                    "l2_peak_GBps": L2_PEAK_GBPS,
                    "l2_frac": requests * 64.0 / launch_s / 1e9 / L2_PEAK_GBPS,
                    "vector_memory_busy_frac": prof.get("TA_BUSY_avr", 0.0) / max(1.0, prof.get("gpu_cycles", 0.0)) or None,
                    "source": provenance})
            valu_issue = None
            if comparable and launch_s > 0:
                # vector instructions of one launch (counter pass) over the live launch time, against the rate at which a SIMD
                # issues wave64 instructions
                peak = SIMDS * CLOCK_HZ / VALU_ARCH_CLOCKS / 1e9
                rate = prof["valu_wave_insts"] / launch_s / 1e9
                valu_issue = {"achieved": rate, "peak": peak, "unit": "G wave-instructions/s", "frac": rate / peak,
                              "frac_of_measured_fast_rate": rate / (SIMDS * CLOCK_HZ / VALU_FAST_CLOCKS / 1e9),
                              "clocks_per_instruction_and_simd": SIMDS * CLOCK_HZ / (rate * 1e9),
                              "wave_insts_per_launch": prof["valu_wave_insts"],
                              "peak_source": "architectural: a wave64 instruction issues over %.0f clocks on a SIMD, 1024 SIMDs, 2.4 GHz; "
                                             "measured on this chip with 8 waves per SIMD (profiles/r03_valu_ceiling.txt): %.2f clocks for fma/mul/add/mov/"
                                             "integer add/logic and for even mixes of those with the other kinds, %.1f for "
                                             "min/max/compare/select/shift/convert on their own"
                                             % (VALU_ARCH_CLOCKS, VALU_FAST_CLOCKS, VALU_SLOW_CLOCKS),
                              "insts_source": provenance}
            # Useful work: the vector lane-operations the algorithm needs (node steps and triangle tests of every lane's own
            # ray, counted by the measurement build of the library in a depth-0-only render) over the lane-operations the chip
            # could have issued in the launch time.  Idle lanes of a wave, bookkeeping and waiting are what is missing to 1.
            useful = None
            want_lanes = args.lane_utilisation != 0 and world == 1
            if want_lanes and launch_s > 0:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import lane_utilisation
                first = lane_utilisation.measure(args.workload, 1, args.grid, args.atlas)         # RECURSION 1: the depth-0 kernel alone
                resident = args.workload == "cubes" and args.resident != 0
                n_node = INSTS_NODE_RESIDENT if resident else INSTS_NODE_MEMORY
                scale = (float(st["pixel_samples"]) / ext_n) / float(first["pixel_samples"])      # a launch against the 32 spp counted
                own = lambda key: (first["bvh_extend_" + key]["steps"] + first["bvh_probe_" + key]["steps"]) * scale
                spent = lambda key: (first["bvh_extend_" + key]["lane_slots"] + first["bvh_probe_" + key]["lane_slots"]) * scale
                lane_ops = own("nodes") * n_node + own("tris") * INSTS_TRIANGLE
                slot_ops = spent("nodes") * n_node + spent("tris") * INSTS_TRIANGLE
                peak_ops = launch_s * PEAK_LANE_TOPS * 1e12
                useful = {"traversal_lane_ops_per_launch": lane_ops, "traversal_frac": lane_ops / peak_ops,
                          "traversal_lane_slots_frac": slot_ops / peak_ops,
                          "traversal_lane_utilisation": lane_ops / max(1.0, slot_ops),
                          "instructions_per_node_step": n_node, "instructions_per_triangle_test": INSTS_TRIANGLE,
                          "source": "node steps and triangle tests of a RECURSION 1 render of 32 spp with gpuraytracer_amd/"
                                    "libgpurt_steps.so (every lane's own steps) and libgpurt_wavesteps.so (lane slots spent), "
                                    "scaled to the samples of a launch; instruction counts from the assembly listing"}
                if valu_issue is not None:
                    # what the launch issued beyond the traversal loops (shading, random numbers, queue records): counted as
                    # useful at full waves, which it nearly is at depth 0 (an upper bound)
                    other = max(0.0, prof["valu_wave_insts"] - slot_ops / 64.0) * 64.0
                    useful["other_lane_ops_per_launch_upper"] = other
                    useful["with_shading_upper_frac"] = (lane_ops + other) / peak_ops
            frac = useful["with_shading_upper_frac"] if useful and "with_shading_upper_frac" in useful else (useful["traversal_frac"] if useful else None)
            # (the 1 M-triangle workload keeps the memory side as its headline: its kernel waits for records, useful.* is beside it)
            if frac is not None and args.workload != "c3":
                roof = {"bound": "valu", "kernel": kernel_name,
                        "achieved": frac * PEAK_LANE_TOPS, "peak": PEAK_LANE_TOPS, "unit": "TFLOP/s", "frac": frac,
                        "traffic": traffic,
                        "traffic_source": provenance if comparable else "not measured for this configuration (%s)" % provenance,
                        "note": "useful vector lane-operations per second (one per lane and instruction: the hit arithmetic may "
                                "not fuse multiply-adds) against 1024 SIMDs x 32 lanes x 2.4 GHz; `frac` counts everything "
                                "outside the traversal loops as useful (upper bound), useful.traversal_frac counts the "
                                "traversal alone; the scene's hierarchy is in LDS, HBM only carries the path records "
                                "(memory_path)" if args.workload == "cubes" else
                                "useful vector lane-operations per second against 1024 SIMDs x 32 lanes x 2.4 GHz (upper bound: "
                                "everything outside the traversal loops counted as useful); the hierarchy is read through L1/L2 "
                                "(memory_path)"}
            else:
                roof = {"bound": "hbm", "kernel": kernel_name,
                        "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                        "traffic": traffic,
                        "traffic_source": provenance if comparable else "not measured for this configuration (%s)" % provenance,
                        "note": "algorithmic bytes against HBM; the kernel waits for node and leaf records that miss LDS and L1 "
                                "(memory_path: L2 rate and hit rate, vector memory busy; valu_issue: what it issues; useful: the share "
                                "of the chip's lane operations that is traversal work of the rays themselves)" if args.workload == "c3" else
                                "traversal step counts were not collected in this run (--lane-utilisation 0, another target size "
                                "or more than one rank): algorithmic bytes against HBM, which is not what bounds this kernel; the "
                                "default single-GPU line reports bound valu with the useful-work fraction"}
            roof["bytes_per_launch"] = algo / ext_n
            roof["memory_path"] = memory_path
            if valu_issue is not None:
                roof["valu_issue"] = valu_issue
            if useful is not None:
                roof["useful"] = useful
        else:
            flops = float(st["pixel_samples"]) * len(scene[0]) * FLOPS_PER_CLOSEST_TEST
            achieved = flops / (ext_ms * 1e-3) / 1e12 if ext_ms > 0 else 0.0
            roof = {"bound": "valu", "kernel": "k_first (depth 0 with loops over all triangles)",
                    "achieved": achieved, "peak": PEAK_VALU_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_VALU_TFLOPS, "traffic": None, "flops_per_launch": flops / ext_n,
                    "note": "closest-hit tests only; the launch also runs the probe loops of depth 0"}
        roof.update({"avg_launch_ms": ext_ms / ext_n, "launches": int(st["extend0_launches"]),
                     "probe_kernels_ms": st["probe_ms"], "extend_kernels_ms": st["extend_ms"],
                     "shade_kernels_ms": st["shade_ms"]})
        spp_label = ("%d spp split over %d GPUs" % (total_spp, world)) if strong and world > 1 else "%d spp per GPU" % (steps * spp)
        out = {
            "metric": "Mrays/s (paths x bounces) at 1080p" if (W, H) == (1920, 1080) else "Mrays/s (paths x bounces) at %dx%d" % (W, H),
            "value": rays / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": dt / max(1, steps) * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s, %dx%d, RECURSION %d, %s (%d spp per step and GPU in passes of %d), constant environment"
                                   % (workload_name, W, H, RECURSION, spp_label, spp, min(args.batch_samples, spp)),
                       "pipeline": "staged" if args.pipeline else "single-kernel",
                       "closest_hit": "hierarchy (exact)" if used_bvh else "loop over all triangles",
                       "parallelism": ("spp split x%d, sums gathered over RCCL and added in rank order" % world) if world > 1 else "1 GPU"},
            "seconds_to_%dspp" % total_spp: dt,
            "seconds_per_1024spp": dt * 1024.0 / float(total_spp),
            "reduce_ms": reduce_ms if dist is not None else 0.0,
            "probe_rays_per_s_M": probes / dt / 1e6,
            "rays_per_pixel_sample": rays / max(1.0, float(st["pixel_samples"]) * world),
            "mean_radiance": mean,
            "nonfinite_pixels": nonfinite_pixels,
            "queue_memory_bytes": int(st["queue_bytes"]),
            "scene_upload_s": upload_s,
            "roofline": roof,
        }
        if world == 1 and used_bvh and not args.no_bruteforce_pass and args.workload == "cubes":
            # the reference's own algorithm (loops over all COUNT triangles per query) on the same GPU: a short
            # pass of depth 0 alone (RECURSION 1), so that every closest-hit loop and every probe loop runs in the
            # one kernel that is timed
            ctx.set_option("accel", 0)
            ctx.set_option("batch_samples", 4)
            ctx.set_frame(cam, clock, 1, 1.0)
            sums.zero_()
            ctx.reset_stats()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ctx.trace(0, 64, 1)
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
            s1 = ctx.stats()
            n_tri = len(scene[0])
            f1 = (float(s1["pixel_samples"]) * n_tri * FLOPS_PER_CLOSEST_TEST
                  + float(s1["shadow_probes"] + s1["glossy_probes"]) * (n_tri - 1) * FLOPS_PER_PROBE_TEST)
            a1 = f1 / (s1["extend0_ms"] * 1e-3) / 1e12 if s1["extend0_ms"] > 0 else 0.0
            out["roofline_bruteforce"] = {
                "bound": "valu", "kernel": "k_first (depth 0 with loops over all %d triangles: closest hit, "
                                           "shading, light probe)" % n_tri,
                "achieved": a1, "peak": PEAK_VALU_TFLOPS, "unit": "TFLOP/s", "frac": a1 / PEAK_VALU_TFLOPS,
                "avg_launch_ms": s1["extend0_ms"] / max(1, s1["extend0_launches"]),
                "flops_per_launch": f1 / max(1, s1["extend0_launches"]),
                "value_Mrays_per_s": s1["rays"] / dt1 / 1e6, "sample": "64 spp, RECURSION 1",
                "note": "flops counted as full tests; triangles whose plane test fails for the whole wave skip the rest"}
        # node visits and triangle tests per query and the lane utilisation of the traversal loops: two 32 spp renders with
        # the measurement builds of the library in child processes (the product does not count its traversal steps)
        want_lanes = args.lane_utilisation != 0
        if want_lanes and world == 1 and args.workload in ("cubes", "obiwan"):
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import lane_utilisation
            out["lane_utilisation"] = lane_utilisation.measure(args.workload, RECURSION)
        if args.no_cpu_baseline or world != 1:
            pass
        elif args.workload == "c3":
            # the CPU restatement loops over all triangles per query as the reference does: a 1080p frame of this scene takes
            # minutes, so the bounded sample is the same camera on a target of 1/12 the width and height
            out["cpu_baseline"] = cpu_baseline(scene, cam, clock, RECURSION, size=(W // 12, H // 12), layers=cpu_layers, atlas=args.atlas)
        else:
            out["cpu_baseline"] = cpu_baseline(scene, cam, clock, RECURSION)
        if args.workload == "cubes":
            out["cpu_baseline_llvmpipe"] = llvmpipe_baseline()
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()

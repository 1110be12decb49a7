#!/usr/bin/env python3
"""bench.py — scheduling decisions/s of the MI355X node-selection engine.

A "step" is one whole NodeSelect cycle (SchedulerAlgo::NodeSelect, reference
src/CraneCtld/JobScheduler.cpp:6507-6836, bracket :1439-1447) over the synthetic C4 queue:
1 M pending jobs x 64 k nodes in 8 disjoint partitions, CPU+mem+GRES requests,
with the job table and node snapshot already resident in HBM when the timed region starts.
At N > 1 the queue is job-sharded by partition (rank r owns partitions p % N == r; partitions that share nodes
stay together; a rank's snapshot lists only its own partitions) and each step ends with one RCCL all-gather of the packed
placement buffers; total work is fixed ("strong").  A partition is ONE sequential chain; k_wide spreads its per-job node scan
over 16 more workgroups of the SAME XCD (exchange through that XCD's L2, ~0.4 us per job), so C4's 8 partitions occupy 136 of
256 CUs and run concurrently on one GPU: more GPUs do not add chains there and the curve is flat by construction (an exchange
over xGMI would cost more per job than the whole chain does now).  --config C4p64 / C4p256 (the same cluster cut
into 64 / 256 partitions) are the layouts on which more GPUs DO add chains: the engine sizes its launch by the partitions that
have pending jobs on the rank, so fewer partitions per rank get the wider k_wide build.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 3 --warmup 1

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (k_wide; k_pipe / k_select when CNS_SELECT_KERNEL=pipe / legacy) with the ALGORITHMIC
bytes of SURVEY.md §8(d) (N_p*S_node + S_job + S_out per decision) over its HIP-event duration;
`cpu_baseline` times the CPU oracle (kind "port": a restatement of the reference algorithm on bit masks, pinned to the
reference's own code by tests/test_ref_pin.py) on ONE WHOLE PARTITION of the same queue (partitions never interact), single
pinned thread like the reference, and diffs its placements against the engine's; `cpu_baseline.reference_build` times THE
REFERENCE'S OWN CODE (oracle/_ref: slices of JobScheduler.{h,cpp} / PublicHeader.{h,cpp} compiled in the build container and
shipped prebuilt) on a bounded prefix of that partition's queue (its whole-partition figures: profiles/r03_ref_vs_oracle_fullsize.txt).
--config C4r / C2r / C5r: the same queue on a cluster that already runs jobs (the cycle CraneCtld normally executes).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def _wide_shape(kernel: str) -> str:
    """'k_wide<2> x64' -> the workgroup shape of that build (scanner waves / 4 scanner workgroups + the home workgroup)."""
    try:
        waves = int(kernel.split(" + ")[0].rsplit(" x", 1)[1].split()[0])
    except (IndexError, ValueError):
        return "single GPU, k_wide"
    return f"single GPU, 1 + {waves // 4} workgroups per partition (k_wide, {waves} scanner waves)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C4")
    ap.add_argument("--jobs", type=int, default=None, help="override J (debug only; invalid as a headline)")
    ap.add_argument("--nodes", type=int, default=None, help="override N (debug only)")
    ap.add_argument("--cpu-sample-jobs", type=int, default=0, help="CPU baseline on the first N jobs of the queue only (0 = whole queue; debug)")
    ap.add_argument("--cpu-partition", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    # --config c5deep: BASELINE config 5's queue (1 M jobs with walltimes in 256 slots of 675 s) on a QUARTER of its nodes.  The frozen C5 (3.75 M
    # cores of demand on 4.19 M) never fills its cluster — every job starts now, nothing is ever packed into a window —; on 16 384 nodes 72 % of the
    # jobs are backfilled: the walltime-packing regime the configuration is named after (tests/golden/make_fullrun.py "c5deep": digest vs the oracle).
    label = args.config
    if args.config.lower() == "c5deep":
        args.config, label = "C5", "c5deep = C5's queue on 16 384 nodes"
        if args.nodes is None:
            args.nodes = 16384

    import torch
    import torch.distributed as dist
    from cranesched_amd import sharding, synth
    from cranesched_amd.engine import GpuNodeSelector

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X; the engine has no CPU fallback"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # CNS_BENCH_FORCE_DIST=1 runs the multi-GPU code path (RCCL init, all-gather of the packed placements) on one
    # rank too — a self-test of that path on a 1-GPU box; the reported line is then not a headline number
    use_dist = world > 1 or os.environ.get("CNS_BENCH_FORCE_DIST") == "1"
    json_fd = 1
    if use_dist:
        # RCCL writes a version banner ("RCCL version : …", "HIP version : …", …) to the C-level stdout of every process that brings a
        # communicator up — buffered, so it lands BEHIND whatever Python printed, whatever NCCL_DEBUG says (seen on ROCm 7.0.2 / RCCL
        # 2.26.6).  The one JSON line must be the only thing on stdout: this process's stdout becomes its stderr from here on, and
        # rank 0 writes the line to the saved descriptor.
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # --config: a frozen queue (C1..C5, C4p64) or its loaded-cluster variant (C4r, C2r, C5r: the same queue on a cluster
    # that already runs jobs, synth.make_running — the cycle CraneCtld normally executes)
    running = None
    base_cfg = synth.LOADED.get(args.config, (args.config,))[0]
    if args.config in synth.LOADED:
        cluster, jobs, now, running = synth.make_loaded(args.config, J=args.jobs, N=args.nodes)
    else:
        cluster, jobs, now = synth.make_config(args.config, J=args.jobs, N=args.nodes)
    # a rank's snapshot lists only ITS partitions (node indices stay global): time maps, costs and the launch — hence the k_wide
    # build, sized by the partitions that have pending jobs — are those of the shard (sharding.shard_cluster)
    my_cluster, my_jobs, my_idx = sharding.shard_cluster(cluster, jobs, rank, world) if world > 1 else (cluster, jobs, np.arange(jobs.num_jobs))

    eng = GpuNodeSelector(device=local_rank)
    eng.set_nodes(my_cluster)
    if running is not None:            # the running jobs on this rank's nodes (every job of make_running lives inside one partition)
        eng.set_running(running if world == 1 else synth.running_of_partitions(
            cluster, running, sharding.partition_plan(cluster.num_partitions, world, sharding.partition_groups(cluster))[rank]))
    eng.upload_jobs(my_jobs)           # inputs resident in HBM before the timed region
    h2d_ms = eng.timing()["h2d_ms"]

    # C4's "+ per-account/QoS limits": the run-limit admission of the commit loop over the resident
    # NodeSelect results.  It runs AFTER the reference's NodeSelect bracket (JobScheduler.cpp:1439-1447 vs :1492-1573),
    # so it is timed separately and reported in "run_limits" next to the headline; single-GPU only (its order is the
    # global pending order, which spans the partition shards).
    limits = None
    if world == 1 and args.config == "C4":
        limits = synth.make_limits(args.config, cluster, jobs)
        eng.set_run_limits(limits[0])

    # The all-gather of the packed placements is the ENGINE's (cns_comm_init_rank + cns_allgather_results: one in-place ncclAllGather on the
    # engine's stream, csrc/group_host.inc); torch.distributed only ships rank 0's communicator id, and carries the barrier and the
    # max-over-ranks of the timing.  Should the engine's communicator not come up on this node, the same bytes go through
    # torch.distributed's all-gather and the line says so ("allgather").
    gather_in = gather_out = None
    gather_via = None
    if use_dist:
        ptr, nbytes = eng.device_results()
        mx = torch.tensor([(nbytes + 15) & ~15], device=dev, dtype=torch.int64)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        pad = int(mx.item())
        try:
            uid = [GpuNodeSelector.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0, device=dev)
            eng.comm_init_rank(world, rank, uid[0])
            ok = torch.tensor([1], device=dev, dtype=torch.int64)
            gather_via = "engine: ncclAllGather inside libcrane_gpu_nodeselect.so (cns_allgather_results)"
        except Exception as ex:   # noqa: BLE001 — reported in the line, never silent
            ok = torch.tensor([0], device=dev, dtype=torch.int64)
            gather_via = f"torch.distributed fallback ({type(ex).__name__}: {ex})"
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:   # (every rank takes the same path)
            if not gather_via.startswith("torch"):
                gather_via = "torch.distributed fallback (another rank's engine communicator did not come up)"
            src = sharding.device_bytes_tensor(ptr, nbytes, dev)
            gather_in = torch.zeros(pad, dtype=torch.uint8, device=dev)
            gather_out = torch.empty(pad * world, dtype=torch.uint8, device=dev)

    def step():
        eng.run_resident(now)                      # init kernel + persistent selection kernel (synchronous)
        if use_dist:                               # merge per-shard node claims: one all-gather over xGMI
            if gather_out is None:
                eng.allgather_results(pad)
            else:
                gather_in[:nbytes].copy_(src)
                dist.all_gather_into_tensor(gather_out, gather_in)

    for _ in range(args.warmup):
        step()
    sel_ms, init_ms = [], []
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        t = eng.timing()
        sel_ms.append(t["select_ms"])
        init_ms.append(t["init_ms"])
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    tm = eng.timing()
    wide_counters = eng.wide_stats() if eng.last_kernel().startswith("k_wide") and my_cluster.num_partitions == cluster.num_partitions else None
    ordered = torch.tensor([tm["jobs_ordered"]], device=dev, dtype=torch.int64)
    el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(ordered, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    total_jobs = int(ordered.item())

    # N > 1 (or CNS_BENCH_FORCE_DIST=1): what the all-gather delivered is unpacked and merged on rank 0 and compared with
    # ONE engine run over the whole queue on rank 0's GPU (untimed): the collective path is checked, not just exercised
    gather_check = None
    if use_dist and rank == 0:
        torch.cuda.synchronize()
        host = eng.download_gathered(pad * world) if gather_out is None else gather_out.cpu().numpy()
        shards = []
        for rk in range(world):
            sj, sidx = sharding.shard(cluster, jobs, rk, world) if world > 1 else (jobs, np.arange(jobs.num_jobs))   # (partition ids are not in the packed buffer)
            shards.append((sharding.unpack_results(host[rk * pad:(rk + 1) * pad], sj, cluster.wide_cores), sidx))
        merged = sharding.merge(jobs, shards)
        eng1 = GpuNodeSelector(device=local_rank)
        eng1.set_nodes(cluster)
        if running is not None:
            eng1.set_running(running)
        single = eng1.node_select(now, jobs)
        eng1.close()
        gather_check = merged.diff(single) is None
    if use_dist:
        dist.barrier()

    lim_line = None
    if limits is not None:
        eng.upload_limit_jobs(limits[1])       # keys resident before the timed passes
        for _ in range(min(args.warmup, 1)):
            eng.run_limits_resident()          # untimed: first-touch allocation of the item streams
        lt, wall = [], []
        for _ in range(max(args.steps, 1)):
            torch.cuda.synchronize()
            w0 = time.perf_counter()
            eng.run_limits_resident()
            wall.append(1e3 * (time.perf_counter() - w0))
            lt.append(eng.limit_timing())
        lim_reason, lim_adm = eng.download_limits()
        lim_line = {"admit_ms": float(np.mean([x["admit_ms"] for x in lt])), "prep_ms": float(np.mean([x["prep_ms"] for x in lt])),
                    "wall_ms": float(np.mean(wall)), "candidates": int(lt[-1]["candidates"]), "admitted": int(lim_adm),
                    "bracketing_rounds": int(lt[-1]["rounds"]), "ordered_fallback": bool(lt[-1]["ordered_fallback"]),
                    "rejected_by": {lm_name: int(n) for lm_name, n in zip(*np.unique(lim_reason[(lim_reason != 0) & (lim_reason != 255)], return_counts=True))},
                    "tables": "1024 users, 64 accounts (8 roots x 7 children), 4 QoS, account x partition caps (synth.make_limits)"}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_jobs * args.steps / elapsed
        avg_sel_ms = float(np.mean(sel_ms))
        achieved = tm["algorithmic_bytes"] / (avg_sel_ms * 1e-3) / 1e9
        # HBM bytes per launch are NOT measured in this run (PMC passes need rocprofv3 around the process): the figure
        # of the committed PMC passes of this same command is quoted under its own key, with its source file.
        traffic_prof = None
        try:
            prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_hbm.json"))
            if prof and args.config == "C4" and args.jobs is None and args.nodes is None and world == 1:
                pj = json.load(open(os.path.join(ROOT, "profiles", prof[-1])))
                traffic_prof = {"bytes_per_launch": pj["traffic_bytes_per_launch"], "source": "profiles/" + prof[-1],
                                "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around this same command on the committed "
                                       "build; 2 x FETCH_SIZE (gfx950) + WRITE_SIZE, mean per launch of the dominant kernel (tools/gpu_bench.sh)"}
        except (OSError, KeyError):
            pass
        got = eng.download()
        kernel = eng.last_kernel()
        # SURVEY 8(d) prices a configuration with time slots (C5: T = 256, Q = 675 s) at N_p * (16 * W + 8) + S_job + S_out bytes per decision,
        # W = ceil(L / Q) slots in the job's window; the engine's own count (cns_timing::algorithmic_bytes) is the T = 1 form N_p * S_node.
        # The line carries the SURVEY's figure for such a configuration; a fraction above 1 is the reuse the survey announces
        # (the window minimum is read from a register tile / one node block, not from W slot records per node).
        algo_bytes, bytes_model = tm["algorithmic_bytes"], "N_p * S_node + 64 + 16 + 24 * k per decision, S_node = 32 (cpu + mem) or 48 (+ GRES / > 64 cores) (SURVEY 8d)"
        if base_cfg == "C5" and world == 1:
            Q = synth.CONFIGS["C5"]["Q"]
            W = (jobs.time_limit_sec.astype(np.int64) + Q - 1) // Q
            npn = np.diff(cluster.part_offsets.astype(np.int64))[jobs.partition.astype(np.int64)]
            algo_bytes = int((npn * (16 * W + 8) + 64 + 16 + 24 * jobs.node_num.astype(np.int64)).sum())
            bytes_model = f"N_p * (16 * W + 8) + 64 + 16 + 24 * k per decision, W = ceil(L / {Q} s) (mean {float(W.mean()):.2f}) (SURVEY 8d, T > 1); the engine's T = 1 count is {tm['algorithmic_bytes']}"
            achieved = algo_bytes / (avg_sel_ms * 1e-3) / 1e9
        # the chain: a partition's decisions are strictly sequential (JobScheduler.cpp:6743 ff.), partitions run side by side — the time per
        # decision ON the busiest partition's chain is what the kernel's latency-bound pipeline delivers, whatever the byte model says
        per_part = np.bincount(my_jobs.partition[my_jobs.partition < my_cluster.num_partitions].astype(np.int64), minlength=my_cluster.num_partitions)
        chain_us = 1e3 * avg_sel_ms / max(int(per_part.max()), 1)
        # SURVEY.md 8(d) bracket (the reference's own, JobScheduler.cpp:1439-1447): the whole cns_select call from host
        # buffers — H2D of the job arrays, record packing, the kernels, D2H of the placements.  Median of 5 after the
        # warm-up above; reported beside `value` (which, per the bench contract, starts with the inputs resident in HBM).
        incl = None
        if world == 1:
            walls = []
            for _ in range(5):
                torch.cuda.synchronize()
                w0 = time.perf_counter()
                eng.node_select(now, my_jobs)
                walls.append(time.perf_counter() - w0)
            # ... and the same call with the caller's job table and result arrays in page-locked memory (cns_host_alloc), result
            # arrays reused across cycles: what an adapter that packs into such buffers sees (DMA straight from / to them; a fresh
            # pageable result array costs a page fault per 4 KB on top of the runtime's staging copies)
            pj, pout = eng.pinned_jobs(my_jobs), eng.pinned_placements(my_jobs)
            eng.node_select(now, pj, out=pout)
            assert got.diff(pout) is None, "page-locked buffers: result differs"
            pwalls = []
            for _ in range(5):
                torch.cuda.synchronize()
                w0 = time.perf_counter()
                eng.node_select(now, pj, out=pout)
                pwalls.append(time.perf_counter() - w0)
            lt = eng.timing()   # (of the last of those cycles)
            incl = {"decisions_per_s": total_jobs / float(np.median(pwalls)), "ms_per_cycle": 1e3 * float(np.median(pwalls)),
                    "what": "whole cns_select from the caller's host buffers: the host pass over the queue (validation, routing, offsets: "
                            "csrc/jobs_host.inc, on host threads) + H2D job arrays + k_pack_jobs + k_prep_jobs + k_init_nodes + "
                            "selection kernel + D2H placements (SURVEY 8d bracket), median of 5; buffers page-locked (cns_host_alloc) and "
                            "reused across cycles",
                    "last_cycle_ms": {"host_pass+h2d+k_pack_jobs": float(lt["h2d_ms"]), "k_init_nodes+k_prep_jobs": float(lt["init_ms"]),
                                      "selection": float(lt["select_ms"]), "d2h": float(lt["d2h_ms"])},
                    "host_threads": os.environ.get("CNS_HOST_THREADS", "default (up to 16)"),
                    "pageable_fresh_buffers": {"decisions_per_s": total_jobs / float(np.median(walls)), "ms_per_cycle": 1e3 * float(np.median(walls)),
                                               "what": "the same call from pageable numpy arrays, result arrays allocated per call"}}
        r = got.reason[:my_jobs.num_jobs]
        line = {
            "metric": "scheduling decisions/sec at 1M pending x 64k nodes",
            # `value`: inputs resident in HBM when the timed region starts (the bench contract: "if the boundary hands over host buffers, note the
            # PCIe-inclusive rate ... it is never `value`").  SURVEY 8(d)'s metric — the reference's own bracket incl. H2D of the job arrays and D2H of
            # the placements — is `incl_h2d_d2h.decisions_per_s` (1 - 2 % below `value`); both are in every line.
            "value": value, "kernel_resident_decisions_per_s": value, "unit": "decisions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int64", "data": "synthetic",
            "config": {"workload": f"{label}: {jobs.num_jobs} pending jobs x {cluster.num_nodes} nodes, "
                                   f"{cluster.num_partitions} disjoint partitions, CPU+mem+GRES(gpu/npu), FIFO, "
                                   f"seed 0x43524E45^{synth.CONFIGS[base_cfg]['idx']}" +
                                   (f"; loaded cluster: {len(running.end_sec)} running jobs, {len(running.alloc_node)} allocations (synth.make_running)" if running is not None else ""),
                       "jobs": jobs.num_jobs, "nodes": cluster.num_nodes, "partitions": cluster.num_partitions,
                       "sharding": "partition p -> rank p % n_gpus; RCCL all-gather of packed placements" if world > 1 else
                                   (_wide_shape(kernel) if kernel.startswith("k_wide") else "single GPU, one workgroup per partition"),
                       "selection_kernel": kernel,
                       "served_by_retry": "retry after" in kernel,   # (cns_run_resident re-runs a cycle on k_pipe / k_select after a k_wide protocol fault: such a line is not a k_wide number)
                       "rank0_outcome": {"start_now": int((r == 0).sum()), "backfilled": int((r == 1).sum()),
                                         "resource": int((r == 2).sum())},
                       "kernel_ms": {kernel: avg_sel_ms, "k_init_nodes+k_prep_jobs+fill": float(np.mean(init_ms))},
                       # k_wide's always-on protocol counters of the last timed step (who waited for whom; partitions whose workgroups
                       # were NOT all on one XCD: their exchange runs ~2.5x slower — a placement the launch cannot force, only report)
                       "wide_protocol_counters": wide_counters,
                       "h2d_job_table_ms": h2d_ms},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # HBM bytes per launch of the dominant kernel from the PMC counters: the committed passes of this same command (a PMC pass needs
                         # rocprofv3 around the process: not measurable from inside the run); null for a workload no pass was committed for
                         "traffic": traffic_prof["bytes_per_launch"] if traffic_prof else None, "traffic_from_profiles": traffic_prof,
                         # what the kernel really moves: the counter bytes of the committed PMC passes / this run's launch time / peak — the
                         # honest utilisation next to the model fraction above (the kernel is a latency-bound sequential chain, not a streaming one)
                         "hbm_util": (traffic_prof["bytes_per_launch"] / (avg_sel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic_prof else None,
                         "chain_us_per_decision": chain_us,
                         # who paces the pipeline of a partition (k_wide's always-on counters): the leader scanner standing in front of a full decision
                         # ring = the home workgroups (test + commit) are behind; the supervisor finding the ring empty = the scanners' chain is
                         "paced_by": (None if not wide_counters else
                                      ("home workgroups (testers)" if wide_counters["leader_polls_ring_full"] > 0.25 * total_jobs else "the scanners' chain")),
                         "leader_polls_ring_full_per_decision": (wide_counters["leader_polls_ring_full"] / max(total_jobs, 1)) if wide_counters else None,
                         "supervisor_looks_empty_per_decision": (wide_counters["looks_empty"] / max(total_jobs, 1)) if wide_counters else None,
                         "home_workgroups_per_partition": (wide_counters.get("home_workgroups", 0) / max(my_cluster.num_partitions, 1)) if wide_counters else None,
                         "chain_note": f"{int(per_part.max())} decisions on the busiest partition's chain, strictly one after the other; {my_cluster.num_partitions} chains side by side",
                         "kernel": kernel, "algorithmic_bytes_per_launch": algo_bytes, "bytes_model": bytes_model,
                         "reuse_factor": (achieved / HBM_PEAK_GBS) if achieved > HBM_PEAK_GBS else None,
                         "avg_launch_ms": avg_sel_ms,
                         "note": "algorithmic bytes = sum over decisions of N_p*48 + 64 + 16 + 24*k (SURVEY 8d); the node "
                                 "tile is register-resident, so HBM traffic is ~1 % of this: the kernel is bound by the latency "
                                 "of a partition's sequential chain (scan -> cross-CU exchange -> decision -> row update, "
                                 "DESIGN.md 4.2/5.2), not by HBM"},
        }
        if incl is not None:
            line["incl_h2d_d2h"] = incl
        if gather_check is not None:
            line["allgather_merged_identical_to_single_gpu"] = bool(gather_check)
            line["allgather"] = gather_via
            if gather_out is None:
                gms, gbytes = eng.gather_timing()
                line["allgather_ms"] = gms
                line["allgather_bytes_per_rank"] = int(gbytes // max(world, 1))
        if lim_line is not None:
            from cranesched_amd import limits as lm
            lim_line["rejected_by"] = {lm.LIMIT_REASON_STR[int(k)]: v for k, v in lim_line["rejected_by"].items()}
            lim_line["decisions_per_s_with_run_limits"] = total_jobs / ((ms_per_step + lim_line["wall_ms"]) * 1e-3)
            lim_line["note"] = ("run-limit admission of the commit loop over the resident NodeSelect results; outside the "
                                "reference's NodeSelect bracket, hence reported beside `value`, not inside it")
            if not args.no_cpu_baseline:
                from oracle import pyoracle  # the checker, timed as the reported CPU figure of this pass
                c0 = time.perf_counter()
                r_ref, a_ref, _ = pyoracle.run_limits(cluster.gres, limits[0], limits[1], got)
                lim_line["cpu_port_ms"] = 1e3 * (time.perf_counter() - c0)
                lim_line["identical_to_cpu_port"] = bool(np.array_equal(r_ref, lim_reason) and a_ref == lim_adm)
            line["run_limits"] = lim_line
        if world == 1 and not args.no_cpu_baseline:
            # CPU baseline on the SAME queue: partitions never interact, so one partition's whole shard (all of its
            # ~J/P jobs on its N/P nodes: cluster filling AND the loaded / backfill regime) is a bounded, unbiased sample
            # of the full queue's per-decision cost.  Single thread like the reference (JobScheduler.cpp:1322,6742),
            # pinned to one core.  Its placements are then diffed against the engine's: a free full-partition parity check.
            from oracle import pyoracle  # the checker, timed here only as the reported baseline
            p0 = args.cpu_partition
            cj = jobs if not args.cpu_sample_jobs else synth.make_config(base_cfg, J=min(args.cpu_sample_jobs, jobs.num_jobs), N=args.nodes)[1]
            sub, idx = synth.select_partitions(cluster, cj, [p0])   # a prefix of the queue keeps the job indices
            run_p0 = None if running is None else synth.running_of_partitions(cluster, running, [p0])
            core = None
            try:
                core = sorted(os.sched_getaffinity(0))[-1]
                os.sched_setaffinity(0, {core})
            except (AttributeError, OSError):
                pass
            ref = pyoracle.select(cluster, sub, now, running=run_p0)
            # ... and THE REFERENCE'S OWN CODE (oracle/_ref: slices of JobScheduler.{h,cpp} / PublicHeader.{h,cpp} compiled in the
            # build container, shipped prebuilt) on a bounded prefix of that partition's queue, the port timed on the same prefix
            # beside it.  A prefix only reaches the cluster-filling regime (it flatters both): the whole-partition figure of the
            # reference build is in profiles/r03_ref_vs_oracle_fullsize.txt.
            ref_build = None
            if pyoracle.ref_available() and not args.cpu_sample_jobs:
                # the prefix grows until the reference build has worked for ~10 s (its cost per decision rises steeply once the
                # partition fills: a longer prefix is predicted from the last one and not started if it would pass ~30 s)
                n_pre, rb = min(sub.num_jobs, 6000), None
                while True:
                    pre_cfg = synth.make_config(base_cfg, J=int(idx[n_pre - 1]) + 1, N=args.nodes)[1]
                    pre_sub, pre_idx = synth.select_partitions(cluster, pre_cfg, [p0])
                    rb = pyoracle.select(cluster, pre_sub, now, running=run_p0, backend="ref")
                    nxt = min(sub.num_jobs, int(n_pre * 1.5))
                    if rb.seconds >= 10.0 or nxt == n_pre or rb.seconds * (nxt / n_pre) ** 3 > 30.0:
                        break
                    n_pre = nxt
                rp = pyoracle.select(cluster, pre_sub, now, running=run_p0)
                same_pre = rb.placements.diff(rp.placements) is None and bool(np.array_equal(rb.placements.start_sec[:pre_sub.num_jobs], got.start_sec[pre_idx]))
                ref_build = {"value": pre_sub.num_jobs / rb.seconds, "unit": "decisions/s", "cores": 1, "kind": "reference",
                             "sample": f"first {pre_sub.num_jobs} jobs of partition {p0} ({rb.seconds:.1f} s on the same pinned core); "
                                       f"the port on the same prefix: {pre_sub.num_jobs / rp.seconds:.0f} decisions/s ({rp.seconds:.2f} s)",
                             "port_on_same_sample_decisions_per_s": pre_sub.num_jobs / rp.seconds,
                             "identical_to_port_and_engine": bool(same_pre)}
            try:
                os.sched_setaffinity(0, set(range(os.cpu_count() or 1)))
            except (AttributeError, OSError):
                pass
            same = bool(np.array_equal(ref.placements.start_sec[:sub.num_jobs], got.start_sec[idx]) and
                        np.array_equal(ref.placements.reason[:sub.num_jobs], got.reason[idx]))
            if same:
                off = got.place_offsets.astype(np.int64) if got.place_offsets[-1] else np.concatenate([[0], np.cumsum(jobs.node_num.astype(np.int64))])
                so = ref.placements.place_offsets.astype(np.int64)
                k = so[1:] - so[:-1]
                dst = np.repeat(off[idx], k) + (np.arange(so[-1]) - np.repeat(so[:-1], k))
                for f in ("node_idx", "ntasks", "cpu_raw", "mem", "core_lo", "core_hi", "gres"):
                    same = same and bool(np.array_equal(getattr(ref.placements, f)[:so[-1]], getattr(got, f)[dst]))
            rr = ref.placements.reason[:sub.num_jobs]
            line["cpu_baseline"] = {
                "value": sub.num_jobs / ref.seconds, "unit": "decisions/s", "cores": 1, "kind": "port",
                "sample": f"partition {p0} of {cluster.num_partitions} of the same queue in full: all {sub.num_jobs} jobs of that "
                          f"shard on its {int(cluster.part_offsets[p0 + 1] - cluster.part_offsets[p0])} nodes "
                          f"({int((rr == 0).sum())} start now, {int((rr == 1).sum())} backfilled), {ref.seconds:.1f} s on one "
                          f"pinned core (core {core}); partitions are i.i.d., so the whole queue costs "
                          f"~{cluster.num_partitions}x that on one core",
                "whole_queue_core_seconds_estimate": ref.seconds * jobs.num_jobs / max(sub.num_jobs, 1),
                "sample_identical_to_engine": same, "host_cpus": os.cpu_count()}
            if ref_build is not None:
                line["cpu_baseline"]["reference_build"] = ref_build
        if json_fd == 1:
            print(json.dumps(line), flush=True)
        else:
            buf = (json.dumps(line) + "\n").encode()
            while buf:
                buf = buf[os.write(json_fd, buf):]
    eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

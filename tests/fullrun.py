"""Full-run digests of a NodeSelect result (SURVEY.md §8d "first 100 k jobs + full-run hash").

One definition shared by the generator (tests/golden/make_fullrun.py, CPU oracle) and the GPU tests
(tests/test_gpu_fullrun.py, HIP engine): a result is reduced to
  * `chunk_crc[c]`  — CRC32 of everything decided for jobs [c*CHUNK, (c+1)*CHUNK): start, reason and the
                      placement records (node, ntasks, cpu, mem, core ids, GRES slots) of those jobs, so that a
                      mismatch is localised to 10 000 jobs;
  * `sha256`        — of all result arrays + the fp64 cost bit patterns of every (partition, node) slot;
  * `timeline_crc`  — CRC32 of the final time maps of a fixed node sample (every node when N <= 4096);
  * `counts`        — jobs per pending reason.
Nothing here depends on who produced the result.
"""
from __future__ import annotations

import hashlib
import zlib

import numpy as np

CHUNK = 10_000
REC_FIELDS = ("node_idx", "ntasks", "cpu_raw", "mem", "core_lo", "core_hi", "gres")


def timeline_nodes(num_nodes: int) -> np.ndarray:
    step = max(1, num_nodes // 2048)
    return np.arange(0, num_nodes, step, dtype=np.int64)


def _b(a) -> bytes:
    return np.ascontiguousarray(a).view(np.uint8).tobytes()


def digest(pl, costs_u64: np.ndarray, timeline_of, num_nodes: int) -> dict:
    """pl: abi.Placements; costs_u64: cost bit patterns per part-slot; timeline_of(node) -> dict of arrays."""
    t = pl.trimmed()
    J = pl.num_jobs
    off = t["place_offsets"].astype(np.int64)
    nchunk = (J + CHUNK - 1) // CHUNK
    crcs = np.zeros(nchunk, np.uint32)
    for c in range(nchunk):
        lo, hi = c * CHUNK, min(J, (c + 1) * CHUNK)
        v = zlib.crc32(_b(t["start_sec"][lo:hi]))
        v = zlib.crc32(_b(t["reason"][lo:hi]), v)
        for f in REC_FIELDS:
            v = zlib.crc32(_b(t[f][off[lo]:off[hi]]), v)
        crcs[c] = v
    h = hashlib.sha256()
    for f in ("start_sec", "reason", "place_offsets") + REC_FIELDS:
        h.update(_b(t[f]))
    h.update(_b(costs_u64))
    tl = 0
    for n in timeline_nodes(num_nodes):
        m = timeline_of(int(n))
        for f in ("t", "cpu_raw", "mem", "core_lo", "core_hi", "gres"):
            tl = zlib.crc32(_b(m[f]), tl)
    return {"chunk_crc": crcs, "sha256": np.frombuffer(h.digest(), np.uint8).copy(),
            "cost_crc": np.array([zlib.crc32(_b(costs_u64))], np.uint32),
            "timeline_crc": np.array([tl], np.uint32),
            "counts": np.bincount(t["reason"], minlength=8).astype(np.int64)}


def preempt_crc(pairs: np.ndarray, cancelled: np.ndarray) -> np.ndarray:
    """CRC of the cycle's preempted_jobs lists — (pending job, reference) rows ordered by job, push_back order within a job;
    reference = running-table index, or pending-queue index | 2^31 — and of the sorted EnqueuePreemptCancel ids."""
    v = zlib.crc32(_b(np.ascontiguousarray(pairs, np.int64)))
    v = zlib.crc32(_b(np.ascontiguousarray(np.sort(cancelled), np.int64)), v)
    return np.array([v], np.uint32)


def compare(got: dict, ref: dict) -> str | None:
    """None if identical, else a description of the first difference."""
    if len(got["chunk_crc"]) != len(ref["chunk_crc"]):
        return f"chunk count {len(got['chunk_crc'])} vs {len(ref['chunk_crc'])}"
    ne = np.nonzero(got["chunk_crc"] != ref["chunk_crc"])[0]
    if len(ne):
        c = int(ne[0])
        return (f"{len(ne)} of {len(ref['chunk_crc'])} job chunks differ, first = jobs [{c * CHUNK}, {(c + 1) * CHUNK}); "
                f"counts {got['counts'].tolist()} vs {ref['counts'].tolist()}")
    if got["cost_crc"][0] != ref["cost_crc"][0]:
        return "placements identical but fp64 cost bit patterns differ"
    if got["timeline_crc"][0] != ref["timeline_crc"][0]:
        return "placements and costs identical but final time maps differ"
    if not np.array_equal(got["sha256"], ref["sha256"]):
        return "sha256 differs"
    if "preempt_crc" in ref and ("preempt_crc" not in got or got["preempt_crc"][0] != ref["preempt_crc"][0]):
        return "placements, costs and time maps identical but the preempted lists / cancel list differ"
    return None

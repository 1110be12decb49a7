// Kernel parameter block shared by the host side of the C ABI (engine.hip) and the gfx950 kernels
// (select_kernels.hip).  All pointers are HBM device pointers.  Layout in HBM:
//   node tables  : SoA / small AoS records indexed by dense node index n or by partition slot q
//   timelines    : tl[n * tl_cap + i], sorted array form of NodeState::time_avail_res_map
//   job table    : SoA, grouped by partition, queue order preserved inside a partition
//   results      : SoA in the caller's original job order
#pragma once
#include <cstddef>
#include "pq_emul.h"
#include "res_dev.h"

namespace cns {

// The per-partition workgroup: 512 threads = 8 wave64 = 2 waves per SIMD, i.e. 256 VGPRs per lane — the
// register-resident node tile and the worker's state fit without spills (at 1024 threads the 128-VGPR cap
// made the hot loops spill whenever cold code changed).  Wave 0 is the worker, waves 1..7 scan.
// Waves w and w+4 share a SIMD and the issue arbiter favours the older wave, so wave 4 scans in the
// worker's issue gaps; measured on C4 / C5 that still beats leaving it without nodes (kIdleWave = 4:
// 786 vs 742 ms), so every non-worker wave holds nodes (kIdleWave >= kWaves: none).
#ifndef CNS_BLOCK
#define CNS_BLOCK 512
#endif
constexpr int kBlock = CNS_BLOCK;            // 512 (2 waves / SIMD, 256 VGPRs); 768 (3 / SIMD, 168 VGPRs) measured 3 % slower
constexpr int kRed = 16;                     // width of the per-wave result exchange (reduce16): >= kWaves
#ifndef CNS_IDLE_WAVE
#define CNS_IDLE_WAVE 99
#endif
constexpr int kIdleWave = CNS_IDLE_WAVE;     // a wave without nodes (>= kWaves: none); it still follows the protocol
constexpr int kScanWaves = kBlock / 64 - 1 - (kIdleWave < kBlock / 64 ? 1 : 0);
constexpr unsigned kScan = kScanWaves * 64;  // scanner lanes = nodes per tile row
// Tile widths (nodes per scanner lane) the selection kernel is instantiated for; a partition uses the
// smallest width w with kScan * w >= its node count.
#if CNS_BLOCK == 768
#define CNS_NPL_LIST(X) X(1) X(3) X(6) X(12) X(24)
#define CNS_NPL_MAX 24
#elif CNS_IDLE_WAVE < 8
#define CNS_NPL_LIST(X) X(1) X(3) X(11) X(22) X(43)
#define CNS_NPL_MAX 43
#else
#define CNS_NPL_LIST(X) X(1) X(3) X(10) X(19) X(28) X(37)
#define CNS_NPL_MAX 37
#endif
constexpr int kWaves = kBlock / 64;
constexpr u32 kTlCap = 1008;         // >= kAlgoMaxJobNumPerNode - 1 + 2 entries per node
constexpr int kMaxUpd = 32;          // per-job owner updates broadcast through LDS
constexpr int kMultiK = kWaves < 8 ? kWaves : 8;      // node_num handled by the parallel multi-node protocol (one helper wave per node, the worker included)
constexpr int kLdsHeap = 33;         // heap / pick entries kept in LDS when node_num < this
constexpr i64 kInf = INT64_MAX;      // absl::InfiniteFuture()
constexpr u32 kNone = 0xFFFFFFFFu;
#define CNS_MAX_NODE_TYPES_DEV 64

// Per-slot node block in HBM: everything the worker needs about one node in ONE contiguous read.
//   header : time-map length, dense node index, node type, cycle-start res_avail (JobScheduler.h:287,313)
//            and res_total
//   entries: sorted-array form of NodeState::time_avail_res_map (JobScheduler.h:245,291)
struct alignas(16) NodeHdr {
  u32 len, node, type, pad;
  Res avail0;
  Res total;
};
static_assert(sizeof(NodeHdr) == 128 && offsetof(NodeHdr, avail0) == 16 && offsetof(NodeHdr, total) == 72 && offsetof(Res, gres) == 32 &&
              offsetof(Res, c2) == 40, "NodeHdr layout (load_block<false> reads the words of its first 96 + 8 bytes by offset)");
static_assert(sizeof(TlMem) == 48 && sizeof(TlExt) == 16, "time-map records in HBM");
constexpr u64 kBlockStride = sizeof(NodeHdr) + (u64)kTlCap * (sizeof(TlMem) + sizeof(TlExt));   // header, kTlCap TlMem, kTlCap TlExt

// Job record: 64 dwords, lane-striped (lane i of a wave loads dword i: one VGPR while in flight).
// Dwords 0..29 are packed by the host (cns_upload_jobs); dwords 32.. are derived per cycle by the
// job-parallel pre-pass k_prep_jobs, so that the sequential chain only reads them out with v_readlane.
constexpr u32 kJobRecDwords = 64;


struct UpdRec {  // what the owning scanner lane must refresh after a commit
  u32 p, len;
  double cost;
  int fcpu;
  u32 fmem;
  u64 fcnt;
  u32 has_front;   // bit 0: the front summary changed; bit 1: keep the cost (the slot of ANOTHER partition on a shared node)
  u32 pad;
};

enum JobFlags : u32 { kJfExclusive = 1, kJfIncl = 2, kJfExcl = 4, kJfGres = 8,
                      kJfMayPreempt = 16 };   // (set per cycle by k_prep_jobs: the job's qos lists a qos it may preempt, JobScheduler.cpp:6384-6385)

// Preemption tables (include/crane_gpu/preempt.h; null / 0 unless the cycle runs with preemption enabled, which forces
// k_select's general path: TryPreempt_ RELEASES resources, the one thing every fast path excludes).
struct PreNode {   // PreemptSegTree::Node, JobScheduler.h:869-877; children are pool indices (0 = none: index 0 is never a child)
  i64 st, ed;
  u32 ls, rs;
  u32 sat, pad;
  Res res, add_tag, sub_tag;
};
struct PreParams {
  u32 enabled, num_qos;
  const u32* qp_off;       // [num_qos + 1] Qos::preempt lists ...
  const u32* qp;           // ... as qos ids
  const u32* pj_qos;       // [J] by queue index (orig): qos id, qos_priority, priority of the pending job
  const u32* pj_qprio;
  const double* pj_prio;
  u32* pj_rec0;            // [J] first placement record of a job placed in this cycle (written at its commit) ...
  u32* pj_k;               // ... and how many
  i64* pj_end;             // ... and its end time (start is o_start[orig])
  const u32* rn_job;       // [A] running allocation entry d (slot-grouped, as rn_end / rn_res) -> running job index
  const u32* ent_slot;     // [A] ... -> its slot
  uint8_t* ent_gone;       // [A] erased from the node's qos_job_map (JobScheduler.h:653-657)
  const u32* rj_qos;       // [R] running job: qos id, qos_priority, start, end as the cycle sees it, in m_preempting_set_?
  const u32* rj_qprio;
  const i64* rj_start;
  const i64* rj_end;
  const uint8_t* rj_preempting;
  const u32* rj_off;       // [R + 1] entries d of running job r ...
  const u32* rj_ent;
  u32* slot_head;          // [S] newest placement record of this cycle on slot q (kNone: none) ...
  u32* rec_next;           // [places] ... linked through here
  u32* rec_orig;           // [places] the pending job of a placement record
  u32* rec_slot;           // [places] its slot
  uint8_t* rec_gone;       // [places] erased from the node's qos_job_map
  char* pool;              // segment-tree nodes, pool_nodes per partition
  u32 pool_nodes, cand_cap;
  u32* cand;               // [P * cand_cap] candidate references (bit 31: pending job), sorted in place
  u32* chosen;             // [P * cand_cap] preempted_jobs of the job at hand
  u32* out_cnt;            // [1] pairs appended so far
  u32* out;                // [2 * out_cap] (pending job (orig), reference) in push_back order per job
  u32 out_cap;
  u32 literal_tree;        // debug / tests: 1 skip the compressed trees (preempt_dev.inc), run every call node for node; 2 give them room for a handful of records only (either also orders every call's candidates block by block)
};

struct KParams {
  // ---- cluster -------------------------------------------------------------------------
  u32 num_nodes, num_parts, num_slots, num_types;
  u32 tl_cap, max_jobs_per_node;
  u32 wide_cores;          // a node of the snapshot has a core id above 127: the TlExt arrays and the o_c2 / o_c3 planes are live
  u32 wide_inject_stall;   // test hook (CNS_WIDE_INJECT_STALL=<job>): job index + 1 of partition 0 whose exchange the leader scanner of k_wide
                           // never publishes — every other wave's wait then runs out (fault 28) and the cycle is re-run on k_pipe / k_select
  u32 wide_window;         // k_wide, 64-wave build: jobs decided per pool exchange at most (0 / 1: one job per exchange, as in rounds 2-4; up to CNS_WIDE_WIN).  CNS_WIDE_WINDOW=<n>
                           // overrides the default (= CNS_WIDE_WIN) for A/B runs and the parity tests; off whenever wide_inject_stall is set
  i64 now, max_window;
  const u32* part_off;     // [P+1] slot range of each partition
  const u32* slot_node;    // [S]   dense node index of slot q (ascending inside a partition)
  const Res* slot_total;   // [S]   res_total of the slot's node (virtual slots: the reserved share)
  const i64* slot_end;     // [S]   end of the slot's time map (INF, or the reservation's end; JobScheduler.h:301-302,337)
  const uint8_t* slot_type;// [S]   node type id (index into type_total)
  const u32* rv_off;       // [S+1] reservation entries (start, end, res) on a real slot, ascending reservation index
  const i64* rv_start;
  const i64* rv_end;
  const Res* rv_res;
  i64* first_resv;         // [S]   earliest start of a non-expired reservation on the node (INF: none), :6635-6642
  const i64* resv_se;      // [2V]  start, end of reservation v (virtual partition num_real_parts + v)
  u32 num_real_parts;
  u32 wide_tester_opt;     // k_wide's testers overlap the record fetch with the node-block load and commit a task that is next to retire straight from
                           // the registers of its test (on unless CNS_WIDE_TESTER_OPT=0: A/B runs)
  u32 wide_aux;            // k_wide: extra home workgroups per partition in this launch (0 .. the build's CNS_WIDE_AUX_MAX; the host sizes it so that every
                           // workgroup of the launch is still resident at once; CNS_WIDE_AUX=<n> caps it: A/B runs and the parity tests)
  u32 wide_batch_post;     // k_wide's supervisor posts a run of one-node decisions in one pass (on unless CNS_WIDE_BATCH_POST=0: A/B runs)
  const Res* type_total;   // [T]   distinct res_total records
  char* blocks;            // [S]   one NodeBlock per partition slot: NodeHdr + tl_cap TlEntry
  u64 block_stride;        //       bytes per NodeBlock
  double* cost;            // [S]   NodeRater::cost (JobScheduler.h:514)
  int* f_cpu;              // [S]   front (t = now) summary: cpu raw, exact
  u32* f_mem;              // [S]   front mem in MiB, rounded up (conservative)
  u64* f_cnt;              // [S]   front GRES popcount per class, byte g
  // One FUTURE entry of the slot's time map that lies below the front in some component ("dip": a backfilled job, a pending
  // reservation, the end of a reservation's map) — a second NECESSARY condition for "starts now": a job whose window reaches
  // past dip_t must also fit the dip.  Time-map entries are never removed and only shrink within a cycle without preemption,
  // so a recorded dip stays valid when it is stale; the committers keep the earliest one (pipe_commit_node), k_wide's
  // scanners pick it up when they reload their tile (without it a loaded cluster trips over the same reservation again and
  // again: 0.7 flushes per job on C4r).
  u32* dip_t;              // [S]   seconds after `now` (0xFFFFFFFF: none)
  u32* dip_cm;             // [S]   whole cpus, rounded up << 16 | memory in GiB, rounded up (both saturating)
  u32* dip_g;              // [S]   GRES popcount per class as 8 saturating nibbles
  // running allocations grouped by node, input order preserved
  const u32* rn_off;       // [N+1]
  const i64* rn_end;       // [A]
  const Res* rn_res;       // [A]
  // ---- jobs, grouped by partition: 32-dword records (JobRecField in select_kernels.hip) ---------
  const u64* pj_off;       // [P+1]
  const u32* jobrec;       // [Jg * kJobRecDwords]
  const u32* incl_nodes;   // included_nodes lists, CSR by the record's incl_b / incl_e
  const u32* excl_nodes;
  // ---- results ---------------------------------------------------------------------------
  i64* o_start;
  uint8_t* o_reason;
  u32* o_node;
  u32* o_ntasks;
  i64* o_cpu;
  u64* o_mem;
  u64* o_clo;
  u64* o_chi;
  u64* o_gres;
  u64* o_c2;               // core ids 128..191 / 192..255 of the placement records; null unless a node of the snapshot has any
  u64* o_c3;
  // ---- scratch ---------------------------------------------------------------------------
  HeapEnt* heap;           // [S + P] partition p uses [part_off[p] + p, ...) of size n_p + 1
  u32* bf_j;               // [S] backfill cursor per selected node
  UpdRec* g_upd;           // [S] owner updates of selections larger than the LDS list
  u32* fault;              // [4] != 0: an internal invariant failed (code, job, aux, aux)
  u64* prof;               // [P*32] cycle counters (only written by -DCNS_PROF builds)
  char* wide_ctl;          // [P] WideCtl blocks of k_wide (exchange rings + control words), zeroed before every launch
  u32* wide_last;          // [P * 65 536] k_wide with 8 / 16 rows per lane: last task posted on a slot (in LDS for the narrower tiles); zeroed before the launch
  u32 general_only;         // != 0: every job through the general path of k_select (preemption enabled)
  u32 serial_only;          // k_wide: the home workgroup alone runs every job through the sequential protocol, its tester waves scanning the
                            // committed HBM arrays (groups of partitions that share nodes and are wider than k_select's register tile)
  // A cycle may be SPLIT over two launches: the partitions that need k_select (groups of partitions that share nodes; with
  // preemption, the partitions whose pending jobs may preempt) and all the others on k_wide / k_pipe, side by side.  Then
  // workgroup (group) i of a launch serves engine partition part_map[i]; null: i itself.
  const u32* part_map;
  u32 launch_parts, pad_lp;
  // ---- the serial-only mode of k_wide (wide groups of partitions that share nodes): compact per-slot map length, kept by
  // k_init_nodes and commit_selection (the only commit path of that mode), and the slot range of every member partition of a
  // group — a job only ever looks at the slots of its own partition: [tag_off[tag_base[part] + tag], ... + 1) relative to the group
  u32* f_len;              // [S] VALID ONLY for the groups of the serial-only mode: k_init_nodes and commit_selection keep it exact there; the inline
                           // and parallel commits of k_select update sibling slots only (nobody reads it outside that mode)
  const u32* tag_off;
  const u32* tag_base;     // [P_real]
  PreParams pre;
  GresDev gres;
  // ---- partitions that share nodes (null otherwise) ----------------------------------------------------------------
  const u32* slot_block;   // [S] slot whose NodeBlock holds the node's (shared) time map = the node's first slot
  const u32* sib_off;      // [S+1] the other slots of the same node (other partitions of the group) ...
  const u32* sib;          //       ... CSR
  const uint8_t* slot_tag; // [S] which partition of its group a slot belongs to (a job only sees the slots of its own partition)
};

}  // namespace cns

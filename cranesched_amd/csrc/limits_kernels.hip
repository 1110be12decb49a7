// Run-limit admission on the GPU (include/crane_gpu/run_limits.h): the QoS / account / partition post-filter of
// the commit loop, src/CraneCtld/JobScheduler.cpp:1492-1573 -> AccountMetaContainer::CheckAndMallocMetaResource
// (src/CraneCtld/Accounting/AccountMetaContainer.cpp:180-224,891-1124).
//
// Data layout in HBM
//   usage  : ONE array of 128-byte records (16 x i64 "components"), all five usage maps back to back
//            [user_qos | user_part | acct_qos | acct_part | qos], + one `exists` byte per record.
//   limits : ONE array of LimRec (same 16 components + max_cpus_per_user + which GRES entries the limit has),
//            [Qos per-user | Qos per-account | Qos global | PartitionResourceLimit ...].
//   components, in the ORDER THE REFERENCE CHECKS THEM inside CheckTres_/CheckGres_:
//            0 cpu   1 jobs   2 wall   3 mem   4..15 GRES: for each name ascending its total, then its classes
//   so that "first failing check" == lowest failing lane, and CheckGres_'s early `return true` (a requested
//   entry the limit does not have, :1034,1043) == a "stop" lane below the first failing one.
//
// Kernels
//   k_lim_flags / k_lim_scan / k_lim_build : job-parallel.  Candidates (NodeSelect reason "" and not skipped) are
//            compacted in order; each gets a record: its allocation view in component order (ResourceV3::View,
//            PublicHeader.cpp:946-952) and, for up to 16 "slots", which usage record / limit record / checks apply:
//            slot 0 user x qos, 1 user x (account, partition), 2 qos global, 3+2l / 4+2l account of tree level l
//            x qos / x partition.  Slots are keyed by the account's LEVEL, not by its position in the chain, so a
//            usage record is read and written by the same 16 lanes for every job that touches it.
//   k_lim_admit : the ordered admission, ONE wave64.  Lane = (slot group of 16 lanes, component); 4 passes cover
//            the 16 slots.  Per job: load the usage components (the only loads that depend on earlier admissions;
//            limits and the next records are prefetched two jobs ahead), compare against the limits, ballots ->
//            admit or first failing check in the reference's order -> store usage + 1 job.  The chain is serial
//            by definition (every admission changes what the next job sees); see DESIGN.md §6.4 for the plan that
//            replaces it by bounded two-sided iterations over key-sorted segments.
//
// Included by engine.hip (one translation unit, namespace cns).
#pragma once

namespace cns {

constexpr u32 kLimSlots = 16;
constexpr u32 kLimPasses = 4;
enum LimKind : u32 { kLkUserQos = 0, kLkUserPart = 1, kLkQos = 2, kLkAcctQos = 3, kLkAcctPart = 4 };
enum LimQosFlag : u32 { kLqUserJobsUnl = 1, kLqAcctJobsUnl = 2, kLqWallZero = 4, kLqUserTresUnl = 8, kLqAcctTresUnl = 16 };
constexpr u32 kLimNeedExists = 1u << 16;   // slot flag: the usage entry must exist (QosEntryNotFound / PartitionEntryNotFound)
constexpr u32 kLimTresComps = 0xFFF9u;     // components of CheckTres_: cpu, mem, GRES
constexpr u32 kLimNotCandidate = 255;

struct LimRec {
  i64 lim[16];
  i64 cpu_x;   // Qos::max_cpus_per_user for per-user QoS records, INT64_MAX otherwise
  u32 has;     // bit c (4..15): the limit's GresMap holds the entry of component c
  u32 pad;
};
static_assert(sizeof(LimRec) == 144, "LimRec layout");

struct LimLayout {
  u64 class_mask[8];
  uint8_t class_comp[8];
  uint8_t class_name_comp[8];  // component of the class's name total
  u32 num_classes, pad;
};

struct LimParams {
  u64 J;
  u32 Q, Pn;
  u32 base_uq, base_up, base_aq, base_ap, base_g, pad0;
  // job keys, pending-vector order
  const u64* sel; const u32* user; const u32* ua; const u32* account; const u32* qos; const u32* part;
  const i64* tl; const uint8_t* skip;
  // tables
  const u32* acct_parent; const u32* acct_level; const u32* qos_flags;
  const u32* user_part_limit; const u32* acct_part_limit;  // may be null
  const LimRec* lim;
  i64* usage;            // [NR * 16]
  uint8_t* exists;       // [NR]
  // NodeSelect results of the last run
  const uint8_t* o_reason; const u64* place_off; const u32* o_node; const i64* o_cpu; const u64* o_mem; const u64* o_gres;
  // compaction + records
  u32* blk_cnt; u64* blk_off; u64* total;
  i64* rec_add;          // [M * 16]
  u32* rec_c; u32* rec_l; u32* rec_en;   // [M * 16]
  u32* rec_job; u32* rec_meta;           // [M]   job index; passes | level << 8
  u64* item_key;         // [M * 16] usage record of (candidate, slot), NR for an unused slot (sort key of the parallel pass)
  u32 NR, pad1;
  uint8_t* out;          // [J] cns_limit_reason
  u64* admitted;
  LimLayout lay;
};

__device__ __forceinline__ bool lim_candidate(const LimParams& P, u64 i) {
  const u64 s = P.sel ? P.sel[i] : i;
  return P.o_reason[s] == 0 && !(P.skip && P.skip[i]);
}

__global__ __launch_bounds__(256) void k_lim_flags(const LimParams P) {
  __shared__ u32 s_cnt[4];
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  bool c = false;
  if (i < P.J) {
    c = lim_candidate(P, i);
    P.out[i] = c ? 0 : kLimNotCandidate;
  }
  const u64 b = __ballot(c);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = (u32)__popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) P.blk_cnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// exclusive scan of the per-block candidate counts (one workgroup; nb = J / 256 entries)
__global__ __launch_bounds__(1024) void k_lim_scan(const LimParams P, u32 nb) {
  __shared__ u64 s_sum[1024];
  const u32 t = threadIdx.x;
  const u32 per = (nb + 1023) / 1024;
  const u32 lo = t * per, hi = lo + per < nb ? lo + per : nb;
  u64 sum = 0;
  for (u32 b = lo; b < hi; ++b) sum += P.blk_cnt[b];
  s_sum[t] = sum;
  __syncthreads();
  for (u32 d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
    const u64 v = t >= d ? s_sum[t - d] : 0;
    __syncthreads();
    s_sum[t] += v;
    __syncthreads();
  }
  u64 run = s_sum[t] - sum;
  for (u32 b = lo; b < hi; ++b) { P.blk_off[b] = run; run += P.blk_cnt[b]; }
  if (t == 1023) *P.total = s_sum[1023];
}

__global__ __launch_bounds__(256) void k_lim_build(const LimParams P) {
  __shared__ u32 s_cnt[4];
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  const bool c = i < P.J && lim_candidate(P, i);
  const u64 b = __ballot(c);
  const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) s_cnt[w] = (u32)__popcll(b);
  __syncthreads();
  if (!c) return;
  u64 k = P.blk_off[blockIdx.x] + (u64)__popcll(b & ((1ull << l) - 1ull));
  for (u32 x = 0; x < w; ++x) k += s_cnt[x];

  const u64 s = P.sel ? P.sel[i] : i;
  const u32 q = P.qos[i], u = P.user[i], ua = P.ua[i], a0 = P.account[i], part = P.part[i];
  // job.allocated_res.View(): sums over the job's nodes; GRES slots counted per (name, type) class
  i64 cpu = 0;
  u64 mem = 0;
  u32 cc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (u64 r = P.place_off[s]; r < P.place_off[s + 1]; ++r) {
    if (P.o_node[r] == kNone) continue;
    cpu += P.o_cpu[r];
    mem += P.o_mem[r];
    const u64 g = P.o_gres[r];
#pragma unroll
    for (u32 x = 0; x < 8; ++x) cc[x] += (u32)__popcll(g & P.lay.class_mask[x]);
  }
  i64* add = P.rec_add + k * 16;
#pragma unroll
  for (u32 x = 0; x < 16; ++x) add[x] = 0;
  add[0] = cpu; add[1] = 1; add[2] = P.tl[i]; add[3] = (i64)mem;
  for (u32 x = 0; x < P.lay.num_classes; ++x)
    if (cc[x]) {  // ResourceView += DedicatedResourceInNode (PublicHeader.cpp:417-427): total and specified grow together
      add[P.lay.class_comp[x]] += cc[x];
      add[P.lay.class_name_comp[x]] += cc[x];
    }

  u32* sc = P.rec_c + k * 16;
  u32* sl = P.rec_l + k * 16;
  u32* se = P.rec_en + k * 16;
#pragma unroll
  for (u32 x = 0; x < 16; ++x) { sc[x] = kNone; sl[x] = kNone; se[x] = 0; }
  const u32 qf = P.qos_flags[q];
  auto part_slot = [&](u32 slot, u32 rec, u32 pl, bool is_user) {
    // CheckPartitionRunLimitsForEntity_ (:542-670): no limit -> no check; each check only when the QoS does not
    // already cap that dimension
    u32 en = 0;
    if (pl != kNone) {
      en = kLimNeedExists;
      if (qf & (is_user ? kLqUserJobsUnl : kLqAcctJobsUnl)) en |= 1u << 1;
      if (qf & kLqWallZero) en |= 1u << 2;
      if (qf & (is_user ? kLqUserTresUnl : kLqAcctTresUnl)) en |= kLimTresComps;
    }
    sc[slot] = rec;
    sl[slot] = pl == kNone ? kNone : 3 * P.Q + pl;
    se[slot] = en | (is_user ? kLkUserPart : kLkAcctPart) << 24;
  };
  sc[0] = P.base_uq + u * P.Q + q; sl[0] = q; se[0] = 0xFFFFu | kLimNeedExists | kLkUserQos << 24;
  part_slot(1, P.base_up + ua * P.Pn + part, P.user_part_limit ? P.user_part_limit[(u64)ua * P.Pn + part] : kNone, true);
  sc[2] = P.base_g + q; sl[2] = 2 * P.Q + q; se[2] = 0xFFFFu | kLkQos << 24;
  const u32 L = P.acct_level[a0];
  for (u32 a = a0; a != kNone; a = P.acct_parent[a]) {  // job.account_chain
    const u32 lv = P.acct_level[a];
    sc[3 + 2 * lv] = P.base_aq + a * P.Q + q; sl[3 + 2 * lv] = P.Q + q;
    se[3 + 2 * lv] = 0xFFFFu | kLimNeedExists | kLkAcctQos << 24;
    part_slot(4 + 2 * lv, P.base_ap + a * P.Pn + part, P.acct_part_limit ? P.acct_part_limit[(u64)a * P.Pn + part] : kNone, false);
  }
  P.rec_job[k] = (u32)i;
  P.rec_meta[k] = ((4 + 2 * L) / 4 + 1) | L << 8;
  if (P.item_key) {
#pragma unroll
    for (u32 x = 0; x < 16; ++x) P.item_key[k * 16 + x] = sc[x] == kNone ? (u64)P.NR : (u64)sc[x];
  }
}

// ---- the ordered admission ---------------------------------------------------------------------------------
struct LimFields { u32 c[kLimPasses], l[kLimPasses], en[kLimPasses]; i64 add; u32 job, meta; };
struct LimLimits { i64 lim[kLimPasses], cpu_x[kLimPasses]; u32 has[kLimPasses]; };

__device__ __forceinline__ void lim_load_fields(const LimParams& P, u64 k, u32 lane, LimFields& F) {
  const u32 comp = lane & 15, grp = lane >> 4;
#pragma unroll
  for (u32 p = 0; p < kLimPasses; ++p) {
    const u64 x = k * 16 + p * 4 + grp;
    F.c[p] = P.rec_c[x]; F.l[p] = P.rec_l[x]; F.en[p] = P.rec_en[x];
  }
  F.add = P.rec_add[k * 16 + comp];
  F.job = P.rec_job[k];
  F.meta = P.rec_meta[k];
}

__device__ __forceinline__ void lim_load_limits(const LimParams& P, const LimFields& F, u32 lane, LimLimits& X) {
  const u32 comp = lane & 15;
#pragma unroll
  for (u32 p = 0; p < kLimPasses; ++p) {
    X.lim[p] = kInf; X.cpu_x[p] = kInf; X.has[p] = 0;
    if (F.l[p] != kNone) {
      const LimRec* r = P.lim + F.l[p];
      X.lim[p] = r->lim[comp];
      X.has[p] = r->has;
      if (comp == 0) X.cpu_x[p] = r->cpu_x;
    }
  }
}

__global__ __launch_bounds__(64) void k_lim_admit(const LimParams P) {
  const u32 lane = threadIdx.x, comp = lane & 15, grp = lane >> 4;
  const u64 M = *P.total;
  u64 adm = 0;
  if (M) {
    LimFields F0, F1, F2;
    LimLimits X0, X1;
    lim_load_fields(P, 0, lane, F0);
    lim_load_fields(P, M > 1 ? 1 : 0, lane, F1);
    lim_load_limits(P, F0, lane, X0);
    for (u64 k = 0; k < M; ++k) {
      // ---- the loads on the critical path: usage of job k (they see every earlier admission) ----
      i64 cnt[kLimPasses];
      u32 ex[kLimPasses];
      const u32 npass = F0.meta & 0xFF, L = F0.meta >> 8;
#pragma unroll
      for (u32 p = 0; p < kLimPasses; ++p) {
        cnt[p] = 0; ex[p] = 1;
        if (F0.c[p] != kNone) {
          cnt[p] = P.usage[(u64)F0.c[p] * 16 + comp];
          if (comp == 0) ex[p] = P.exists[F0.c[p]];
        }
      }
      // ---- off the critical path: limits of job k+1, records of job k+2 ----
      lim_load_limits(P, F1, lane, X1);
      lim_load_fields(P, k + 2 < M ? k + 2 : M - 1, lane, F2);

      // ---- checks ----
      i64 use[kLimPasses];
      u32 v = 0xFFFFu;          // (order key << 8 | reason) of this lane's first failing check
      u64 any_gres_fail = 0;
      bool gf[kLimPasses];
      u64 sf[kLimPasses];
#pragma unroll
      for (u32 p = 0; p < kLimPasses; ++p) {
        gf[p] = false; sf[p] = 0;
        use[p] = cnt[p] + F0.add;
        if (p >= npass) continue;  // uniform
        const u32 en = F0.en[p];
        const bool on = (en >> comp & 1) != 0 && F0.c[p] != kNone;
        const u32 kind = en >> 24;
        const bool partk = kind == kLkUserPart || kind == kLkAcctPart;
        // reference order of the slots: user x qos, user x partition, the account chain from the job's account
        // up to the root (qos then partition each), the qos globally (CheckRunLimits_ :891-1028)
        const u32 slot = p * 4 + grp;
        const u32 rank = slot == 0 ? 0 : slot == 1 ? 1 : slot == 2 ? 4 + 2 * L : 2 + 2 * (L - ((slot - 3) >> 1)) + ((slot - 3) & 1);
        u32 within = 7;
        bool stop = false;
        if (comp == 0) {
          if ((en & kLimNeedExists) && !ex[p]) within = 0;                       // entry not found
          else if (on && use[p] > X0.cpu_x[p]) within = 1;                      // QosCpuResourceLimit
          else if (on && use[p] > X0.lim[p]) within = 4;                        // (Partition)CpuResourceLimit
        } else if (comp < 4) {
          if (on && use[p] > X0.lim[p]) within = comp == 1 ? 2 : comp == 2 ? 3 : 5;  // jobs, wall, mem
        } else {
          const bool present = use[p] > 0;                                       // the use GresMap has this entry
          const bool has = (X0.has[p] >> comp & 1) != 0;
          stop = on && present && !has;                                          // CheckGres_ `return true`
          gf[p] = on && present && has && use[p] > X0.lim[p];
        }
        sf[p] = __ballot(stop || gf[p]);
        any_gres_fail |= __ballot(gf[p]);
        if (within < 7) {
          const u32 code = !partk ? 1 + within
                                  : within == 0 ? 8 : within == 2 ? (kind == kLkUserPart ? 9 : 11)
                                  : within == 3 ? (kind == kLkUserPart ? 10 : 12) : 9 + within;  // 4,5 -> 13,14
          const u32 cand = (rank * 8 + within) << 8 | code;
          v = cand < v ? cand : v;
        }
      }
      if (any_gres_fail) {  // uniform, rare: a GRES component fails unless a lower one of its slot stopped the walk
#pragma unroll
        for (u32 p = 0; p < kLimPasses; ++p) {
          const u32 below = (u32)(sf[p] >> (lane & ~15u)) & ((1u << comp) - 1u);
          if (gf[p] && below == 0) {
            const u32 en = F0.en[p], kind = en >> 24, slot = p * 4 + grp;
            const bool partk = kind == kLkUserPart || kind == kLkAcctPart;
            const u32 rank = slot == 0 ? 0 : slot == 1 ? 1 : slot == 2 ? 4 + 2 * L : 2 + 2 * (L - ((slot - 3) >> 1)) + ((slot - 3) & 1);
            const u32 cand = (rank * 8 + 6) << 8 | (partk ? 15u : 7u);
            v = cand < v ? cand : v;
          }
        }
      }
      u64 cand = __ballot(v != 0xFFFFu);
      u32 reason = 0;
      if (cand) {
        // first failing check in the reference's order = minimum key: 7 ballots over the key bits
#pragma unroll
        for (int bit = 14; bit >= 8; --bit) {
          const u64 zero = __ballot((v >> bit & 1) == 0) & cand;
          cand = zero ? zero : cand;
        }
        reason = (u32)__builtin_amdgcn_readlane((int)v, (int)__builtin_ctzll(cand)) & 0xFF;
      } else {
        // DoMallocResource_ (:1067-1124): every usage record of the job grows, missing entries are created
#pragma unroll
        for (u32 p = 0; p < kLimPasses; ++p)
          if (p < npass && F0.c[p] != kNone) {
            P.usage[(u64)F0.c[p] * 16 + comp] = use[p];
            if (comp == 0 && !ex[p]) P.exists[F0.c[p]] = 1;
          }
        ++adm;
      }
      if (lane == 0) P.out[F0.job] = (uint8_t)reason;
      F0 = F1; F1 = F2; X0 = X1;
    }
  }
  if (lane == 0) *P.admitted = adm;
}

// ==== the parallel admission ===================================================================================
// Greedy admission in order is a chain, but its DECISIONS can be bracketed: with A = jobs known to be admitted and
// X = jobs known to be rejected, every usage record a job sees lies between usage0 + (sum over earlier jobs in A)
// and usage0 + (sum over earlier jobs not in X).  A job all of whose checks pass over that whole interval is
// admitted whatever the undecided jobs turn out to be; one with a check that fails over the whole interval is
// rejected.  The first undecided job always has a zero-width interval, so every round decides at least one job;
// on the C4 tables ~10 rounds decide all 759 k candidates.  If kLimMaxRounds do not suffice the host falls back to
// the ordered single-wave kernel above (same results by construction, both are checked against the oracle).
//
// "Sum over earlier jobs with the same usage record" = segmented exclusive prefix sums over the (candidate, slot)
// items sorted by usage record (stable LSD radix sort, once per cycle; order inside a record = candidate order).
// The sorted items are gathered once into streams (candidate, limit, checks, 16 components) so that a round reads
// HBM linearly: per round k_par_tails (partial sums per chunk), k_par_carry (scan over the chunk tails),
// k_par_eval (re-walk with the carry, evaluate the undecided jobs' slots) and k_par_update.
// The CheckGres_ walk is not monotone in the usage (an entry the limit lacks stops the walk, :1034,1043): its
// outcome over an interval is evaluated in three-valued logic (certain stop / maybe stop / certain fail / maybe fail).
constexpr u32 kParChunks = 8192;     // chunks of the sorted item stream (16 lanes walk one chunk)
constexpr u32 kParMinChunk = 64;
constexpr u32 kParBatch = 8;         // items loaded together by a chunk walker
constexpr u32 kLimMaxRounds = 64;

struct ParParams {
  const u64* n_items;     // items with a usage record (the sorted stream's prefix)
  const u64* total;       // candidates
  const u32* s_key;       // [n] usage record of the item
  const u32* s_k;         // [n] candidate
  const u32* s_l;         // [n] limit record or kNone
  const u32* s_en;        // [n] checks of the slot (kind << 24 | need-exists | component mask)
  const u32* s_slot;      // [n] slot number (for the reason order)
  const i64* s_add;       // [n * 16]
  uint8_t* state;         // [M] 0 undecided, 1 admitted, 2 rejected
  u32* flags;             // [M] per round: bit 0 a slot certainly fails, bit 1 a slot does not certainly pass
  u32* jobkey;            // [M] final pass: min (order key << 8 | reason) over the failing checks
  i64* tails;             // [chunks][2][16]
  uint8_t* heads;         // [chunks]
  i64* carry;             // [chunks][2][16]
  u64* undecided;
  const LimRec* lim;
  const i64* usage0; const uint8_t* exists0;
  i64* usage; uint8_t* exists;
  const u32* rec_meta; const u32* rec_job;
  uint8_t* out; u64* admitted;
};

__device__ __forceinline__ u64 par_chunk_len(u64 n) {
  const u64 c = (n + kParChunks - 1) / kParChunks;
  return c < kParMinChunk ? kParMinChunk : c;
}

// sorted (key, item) pairs -> streams; also finds how many items carry a usage record
__global__ __launch_bounds__(256) void k_par_gather(const u64* __restrict__ keys, const u32* __restrict__ vals, u64 n, u32 NR,
                                                    const LimParams L, u32* s_key, u32* s_k, u32* s_l, u32* s_en, u32* s_slot,
                                                    i64* s_add, u64* n_items) {
  const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
  const u64 i = t >> 4;
  const u32 comp = (u32)t & 15;
  if (i >= n) return;
  const u64 key = keys[i];
  if (key >= NR) return;
  const u32 it = vals[i], k = it >> 4;
  if (comp == 0) {
    s_key[i] = (u32)key; s_k[i] = k; s_l[i] = L.rec_l[it]; s_en[i] = L.rec_en[it]; s_slot[i] = it & 15;
    if (i + 1 == n || keys[i + 1] >= NR) *n_items = i + 1;
  }
  s_add[i * 16 + comp] = L.rec_add[(u64)k * 16 + comp];
}

// one 16-lane group per chunk: sums of the last segment of the chunk, over A (admitted) and over not-X
__global__ __launch_bounds__(256) void k_par_tails(const ParParams P) {
  const u32 g = (blockIdx.x * 256 + threadIdx.x) >> 4, comp = threadIdx.x & 15;
  const u64 n = *P.n_items, C = par_chunk_len(n);
  const u64 beg = (u64)g * C, end = beg + C < n ? beg + C : n;
  i64 aL = 0, aU = 0;
  bool head = false;
  if (beg < end) {
    u32 prev = beg ? P.s_key[beg - 1] : kNone;
    for (u64 i = beg; i < end; i += kParBatch) {
      u32 key[kParBatch], k[kParBatch], st[kParBatch];
      i64 add[kParBatch];
#pragma unroll
      for (u32 b = 0; b < kParBatch; ++b) {
        const u64 x = i + b < end ? i + b : end - 1;
        key[b] = P.s_key[x]; k[b] = P.s_k[x]; add[b] = P.s_add[x * 16 + comp];
      }
#pragma unroll
      for (u32 b = 0; b < kParBatch; ++b) st[b] = P.state[k[b]];
#pragma unroll
      for (u32 b = 0; b < kParBatch; ++b)
        if (i + b < end) {
          if (key[b] != prev) { aL = 0; aU = 0; head = true; prev = key[b]; }
          aL += st[b] == 1 ? add[b] : 0;
          aU += st[b] != 2 ? add[b] : 0;
        }
    }
  }
  P.tails[((u64)g * 2 + 0) * 16 + comp] = aL;
  P.tails[((u64)g * 2 + 1) * 16 + comp] = aU;
  if (comp == 0) P.heads[g] = head;
}

// carry[c] = sum of the items since the last segment head before chunk c:  x[c] = heads[c-1] ? tails[c-1] : x[c-1] + tails[c-1]
// one workgroup: 64 row groups of 16 lanes, each over kParChunks / 64 consecutive chunks, two sweeps
__global__ __launch_bounds__(1024) void k_par_carry(const ParParams P) {
  __shared__ i64 s_end[64][2][16];
  __shared__ i64 s_in[64][2][16];
  __shared__ uint8_t s_seen[64];
  const u32 rg = threadIdx.x >> 4, comp = threadIdx.x & 15;
  constexpr u32 R = kParChunks / 64;
  const u32 c0 = rg * R;
  for (u32 sweep = 0; sweep < 2; ++sweep) {
    i64 xL = sweep ? s_in[rg][0][comp] : 0, xU = sweep ? s_in[rg][1][comp] : 0;
    bool seen = false;
    for (u32 c = c0; c < c0 + R; ++c) {
      if (sweep) { P.carry[((u64)c * 2 + 0) * 16 + comp] = xL; P.carry[((u64)c * 2 + 1) * 16 + comp] = xU; }
      const i64 tL = P.tails[((u64)c * 2 + 0) * 16 + comp], tU = P.tails[((u64)c * 2 + 1) * 16 + comp];
      if (P.heads[c]) { xL = tL; xU = tU; seen = true; } else { xL += tL; xU += tU; }
    }
    if (sweep == 0) {
      s_end[rg][0][comp] = xL; s_end[rg][1][comp] = xU;
      if (comp == 0) s_seen[rg] = seen;
      __syncthreads();
      if (rg == 0) {   // 64 sequential steps over the row groups
        i64 yL = 0, yU = 0;
        for (u32 r = 0; r < 64; ++r) {
          s_in[r][0][comp] = yL; s_in[r][1][comp] = yU;
          if (s_seen[r]) { yL = s_end[r][0][comp]; yU = s_end[r][1][comp]; } else { yL += s_end[r][0][comp]; yU += s_end[r][1][comp]; }
        }
      }
      __syncthreads();
    }
  }
}

// first failing check of one slot in the reference's order -> (order key << 8 | reason), 0xFFFF if the slot passes
// (exact usage; used by the final pass).  `grp16` = this group's 16 bits of a wave ballot.
__device__ __forceinline__ u32 par_slot_reason(i64 use, i64 lim, i64 cpu_x, u32 has, u32 en, bool exists, u32 comp, u32 slot, u32 L,
                                               u32 lane) {
  const u32 kind = en >> 24;
  const bool partk = kind == kLkUserPart || kind == kLkAcctPart;
  const bool on = (en >> comp & 1) != 0;
  const u32 rank = slot == 0 ? 0 : slot == 1 ? 1 : slot == 2 ? 4 + 2 * L : 2 + 2 * (L - ((slot - 3) >> 1)) + ((slot - 3) & 1);
  u32 within = 7;
  bool stop = false, gf = false;
  if (comp == 0) {
    if (on && use > cpu_x) within = 1;
    else if (on && use > lim) within = 4;
  } else if (comp == 1) {
    if ((en & kLimNeedExists) && !exists) within = 0;
    else if (on && use > lim) within = 2;
  } else if (comp < 4) {
    if (on && use > lim) within = comp == 2 ? 3 : 5;
  } else {
    const bool present = use > 0, h = (has >> comp & 1) != 0;
    stop = on && present && !h;
    gf = on && present && h && use > lim;
  }
  const u32 sf = (u32)(__ballot(stop || gf) >> (lane & 48u)) & 0xFFFFu;
  if (gf && (sf & ((1u << comp) - 1u)) == 0) within = 6;
  if (within == 7) return 0xFFFFu;
  const u32 code = !partk ? 1 + within
                          : within == 0 ? 8 : within == 2 ? (kind == kLkUserPart ? 9 : 11)
                          : within == 3 ? (kind == kLkUserPart ? 10 : 12) : 9 + within;
  return (rank * 8 + within) << 8 | code;
}

// FINAL = false: bracket the undecided jobs' slots.  FINAL = true: every job is decided, the sums are exact:
// reasons of the rejected jobs, usage tables after the pass.
template <bool FINAL>
__global__ __launch_bounds__(256) void k_par_eval(const ParParams P) {
  const u32 g = (blockIdx.x * 256 + threadIdx.x) >> 4, comp = threadIdx.x & 15, lane = threadIdx.x & 63;
  const u64 n = *P.n_items, C = par_chunk_len(n);
  const u64 beg = (u64)g * C, end = beg + C < n ? beg + C : n;
  const bool live = beg < end;
  i64 aL = live ? P.carry[((u64)g * 2 + 0) * 16 + comp] : 0, aU = live ? P.carry[((u64)g * 2 + 1) * 16 + comp] : 0;
  u32 prev = live && beg ? P.s_key[beg - 1] : kNone;
  const u64 steps = (C + kParBatch - 1) / kParBatch;   // uniform trip count: the ballots below need every lane
  for (u64 t = 0; t < steps; ++t) {
    const u64 i = beg + t * kParBatch;
    // two load stages per batch, each issued for all kParBatch items before anything is consumed: the item streams
    // (usage record, candidate, components, limit, checks), then what they point to (the candidate's state, the limit
    // record, usage0).  Every load is unconditional on a clamped index and masked afterwards: a load under a
    // data-dependent branch makes the compiler wait for it on the spot (one round trip per item instead of per batch).
    u32 key[kParBatch], k[kParBatch], st[kParBatch], lidx[kParBatch], en[kParBatch], slot[kParBatch];
    i64 add[kParBatch];
    bool act[kParBatch], need[kParBatch];
#pragma unroll
    for (u32 b = 0; b < kParBatch; ++b) {
      act[b] = live && i + b < end;
      const u64 x = act[b] ? i + b : 0;   // item 0 always exists (every candidate has at least three slots)
      key[b] = P.s_key[x]; k[b] = P.s_k[x]; add[b] = P.s_add[x * 16 + comp];
      lidx[b] = P.s_l[x]; en[b] = P.s_en[x];
      slot[b] = FINAL ? P.s_slot[x] : 0;
    }
    i64 lim[kParBatch], cpu_x[kParBatch], c0[kParBatch];
    u32 has[kParBatch], lvl[kParBatch];
    bool ex0[kParBatch];
    bool any = false;
#pragma unroll
    for (u32 b = 0; b < kParBatch; ++b) {
      st[b] = P.state[k[b]];
      const LimRec* r = P.lim + (lidx[b] == kNone ? 0u : lidx[b]);
      lim[b] = r->lim[comp]; has[b] = r->has; cpu_x[b] = r->cpu_x;
      c0[b] = P.usage0[(u64)key[b] * 16 + comp];
      ex0[b] = P.exists0[key[b]] != 0;
      lvl[b] = FINAL ? P.rec_meta[k[b]] >> 8 : 0;
    }
#pragma unroll
    for (u32 b = 0; b < kParBatch; ++b) {
      if (lidx[b] == kNone) { lim[b] = kInf; cpu_x[b] = kInf; has[b] = 0; }
      if (comp != 0) cpu_x[b] = kInf;
      if (comp != 1) ex0[b] = true;
      if (!act[b]) { st[b] = 2; key[b] = kNone; add[b] = 0; }
    }
#pragma unroll
    for (u32 b = 0; b < kParBatch; ++b) {
      need[b] = act[b] && (FINAL ? st[b] == 2 : st[b] == 0);
      any = any || need[b];
    }
    const bool wave_any = __ballot(any) != 0;
#pragma unroll
    for (u32 b = 0; b < kParBatch; ++b) {
      if (act[b] && key[b] != prev) {
        if (FINAL && prev != kNone) {  // the previous usage record is complete: DoMallocResource_'s result
          P.usage[(u64)prev * 16 + comp] = P.usage0[(u64)prev * 16 + comp] + aL;
          if (comp == 1 && aL > 0) P.exists[prev] = 1;
        }
        aL = 0; aU = 0; prev = key[b];
      }
      if (wave_any && __ballot(need[b])) {
        const bool nd = need[b];
        const i64 useL = c0[b] + aL + add[b], useU = c0[b] + aU + add[b];
        if (FINAL) {
          const u32 v = par_slot_reason(useL, lim[b], cpu_x[b], has[b], en[b], ex0[b] || aL > 0, comp, slot[b], lvl[b], lane);
          if (nd && v != 0xFFFFu) atomicMin(&P.jobkey[k[b]], v);
        } else {
          const bool on = nd && (en[b] >> comp & 1) != 0;
          bool fC = false, fM = false, sC = false, sM = false, gC = false, gM = false;
          if (comp == 0) {
            fC = on && (useL > cpu_x[b] || useL > lim[b]);
            fM = on && (useU > cpu_x[b] || useU > lim[b]);
          } else if (comp < 4) {
            fC = on && useL > lim[b];
            fM = on && useU > lim[b];
            if (comp == 1 && nd && (en[b] & kLimNeedExists)) {   // the entry exists once any earlier job was admitted
              const bool exL = ex0[b] || aL > 0, exU = ex0[b] || aU > 0;
              fC = fC || !exU;
              fM = fM || !exL;
            }
          } else {
            const bool h = (has[b] >> comp & 1) != 0, pC = useL > 0, pM = useU > 0;
            sC = on && !h && pC;
            sM = on && !h && pM;
            gC = on && h && useL > lim[b];
            gM = on && h && useU > lim[b];
          }
          const u32 sh = lane & 48u;
          const u32 bfC = (u32)(__ballot(fC) >> sh) & 0xFFFFu, bfM = (u32)(__ballot(fM) >> sh) & 0xFFFFu;
          const u64 any_gres = __ballot(sM || gM);
          bool g_pass = true, g_fail = false;
          if (any_gres) {
            const u32 bsC = (u32)(__ballot(sC) >> sh) & 0xFFFFu, bsM = (u32)(__ballot(sM) >> sh) & 0xFFFFu;
            const u32 bgC = (u32)(__ballot(gC) >> sh) & 0xFFFFu, bgM = (u32)(__ballot(gM) >> sh) & 0xFFFFu;
            // CheckGres_ over the interval: certainly passes iff the first of {certain stop, any fail} is a certain
            // stop (or none exists); certainly fails iff the first of {any stop, any fail} is a certain fail
            const u32 m1 = bsC | bgC | bgM, m2 = m1 | bsM;
            g_pass = m1 == 0 || (bsC & (m1 & (0u - m1))) != 0;
            g_fail = m2 != 0 && (bgC & (m2 & (0u - m2))) != 0;
          }
          const bool cfail = bfC != 0 || g_fail;
          const bool cpass = bfM == 0 && g_pass;   // bfM includes bfC
          if (nd && comp == 0 && (cfail || !cpass)) atomicOr(&P.flags[k[b]], (cfail ? 1u : 0u) | (cpass ? 0u : 2u));
        }
      }
      if (act[b]) {
        aL += st[b] == 1 ? add[b] : 0;
        aU += st[b] != 2 ? add[b] : 0;
      }
    }
  }
  if (FINAL && live && end == n && prev != kNone) {   // the very last usage record
    P.usage[(u64)prev * 16 + comp] = P.usage0[(u64)prev * 16 + comp] + aL;
    if (comp == 1 && aL > 0) P.exists[prev] = 1;
  }
}

__global__ __launch_bounds__(256) void k_par_update(const ParParams P) {
  const u64 k = (u64)blockIdx.x * 256 + threadIdx.x;
  bool und = false;
  if (k < *P.total && P.state[k] == 0) {
    const u32 f = P.flags[k];
    P.flags[k] = 0;
    if (f & 1) P.state[k] = 2;
    else if (!(f & 2)) P.state[k] = 1;
    else und = true;
  }
  const u64 b = __ballot(und);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd((unsigned long long*)P.undecided, (unsigned long long)__popcll(b));
}

__global__ __launch_bounds__(256) void k_par_finish(const ParParams P) {
  const u64 k = (u64)blockIdx.x * 256 + threadIdx.x;
  bool adm = false;
  if (k < *P.total) {
    adm = P.state[k] == 1;
    P.out[P.rec_job[k]] = adm ? 0 : (uint8_t)(P.jobkey[k] & 0xFF);
  }
  const u64 b = __ballot(adm);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd((unsigned long long*)P.admitted, (unsigned long long)__popcll(b));
}

}  // namespace cns

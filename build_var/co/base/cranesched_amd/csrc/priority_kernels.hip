// MultiFactorPriority on the GPU: bounds (min/max reductions) -> per-account service values -> one fp64
// priority per pending job -> stable LSD radix sort by descending priority.
//
// Reference: src/CraneCtld/JobScheduler.cpp:7606-7819 (see include/crane_gpu/priority.h).  Unlike the node
// selection chain this path is embarrassingly parallel and HBM-bound: every kernel streams SoA arrays with
// coalesced loads; reductions go wave (DPP) -> one atomic per wave.  fp64 is evaluated in the reference's
// operation order with -ffp-contract=off, so priorities match the CPU bit for bit.
//
// Included by engine.hip after select_kernels.hip (one translation unit, namespace cns).
#pragma once

namespace cns {

struct PrioBounds {  // FactorBound, JobScheduler.h:214-224 (+ the per-account map as two arrays)
  u64 age_max, age_min;
  u64 mem_max, mem_min;
  u64 cpus_max_bits, cpus_min_bits;  // non-negative doubles order like their bit patterns
  double sv_max, sv_min;
  u32 qos_max, qos_min;
  u32 part_max, part_min;
  u32 nn_max, nn_min;
};

struct PrioParams {
  i64 now;
  u64 max_age;
  u32 w_age, w_fair, w_size, w_part, w_qos, favor_small;
  u32 J, R, A, pad;
  // pending
  const i64* submit; const u32* qos; const u32* part; const u32* node_num; const i64* cpu_raw; const u64* mem;
  const u32* account; const double* cached;
  // running
  const i64* r_start; const u32* r_qos; const u32* r_part; const u32* r_node_num; const i64* r_cpu_raw;
  const u64* r_mem; const u32* r_account;
  const u32* acc_off;     // [A+1] CSR: running jobs of each account, in vector order
  const u32* acc_jobs;    // [R]
  const uint8_t* acc_present;  // [A] the account appears among the pending or running jobs
  PrioBounds* bounds;
  double* acc_val;        // [A]
  double* terms;          // [R] service_val * run_time of the running jobs, in account-CSR order
  double* prio;           // [J]
  u64* keys;              // [J] sort keys (ascending = descending priority)
};

__device__ __forceinline__ double prio_cpu_double(i64 raw) { return (double)raw / 256.0; }  // PublicHeader.cpp:509-511

// k_prio_bounds: one pass over the pending and the running jobs (CalculateFactorBound_ :7663-7713).
__global__ __launch_bounds__(256) void k_prio_bounds(const PrioParams P) {
  u64 age_max = 0, age_min = ~0ull, mem_max = 0, mem_min = ~0ull, c_max = 0, c_min = ~0ull;
  u32 qos_max = 0, qos_min = ~0u, part_max = 0, part_min = ~0u, nn_max = 0, nn_min = ~0u;
  const u32 stride = gridDim.x * blockDim.x;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < P.J; i += stride) {
    u64 age = (u64)(P.now - P.submit[i]);
    age = age < P.max_age ? age : P.max_age;
    age_max = age > age_max ? age : age_max; age_min = age < age_min ? age : age_min;
    const u64 m = P.mem[i];
    mem_max = m > mem_max ? m : mem_max; mem_min = m < mem_min ? m : mem_min;
    const u64 cb = (u64)__double_as_longlong(prio_cpu_double(P.cpu_raw[i]));
    c_max = cb > c_max ? cb : c_max; c_min = cb < c_min ? cb : c_min;
    const u32 q = P.qos[i], pp = P.part[i], nn = P.node_num[i];
    qos_max = q > qos_max ? q : qos_max; qos_min = q < qos_min ? q : qos_min;
    part_max = pp > part_max ? pp : part_max; part_min = pp < part_min ? pp : part_min;
    nn_max = nn > nn_max ? nn : nn_max; nn_min = nn < nn_min ? nn : nn_min;
  }
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < P.R; i += stride) {
    const u64 m = P.r_mem[i];
    mem_max = m > mem_max ? m : mem_max; mem_min = m < mem_min ? m : mem_min;
    const u64 cb = (u64)__double_as_longlong(prio_cpu_double(P.r_cpu_raw[i]));
    c_max = cb > c_max ? cb : c_max; c_min = cb < c_min ? cb : c_min;
    const u32 q = P.r_qos[i], pp = P.r_part[i], nn = P.r_node_num[i];
    qos_max = q > qos_max ? q : qos_max; qos_min = q < qos_min ? q : qos_min;
    part_max = pp > part_max ? pp : part_max; part_min = pp < part_min ? pp : part_min;
    nn_max = nn > nn_max ? nn : nn_max; nn_min = nn < nn_min ? nn : nn_min;
  }
  // wave reduction on the DPP crossbar -> block fold through LDS -> one set of atomics per block
  age_max = wave_max_u64(age_max); age_min = wave_min_u64(age_min);
  mem_max = wave_max_u64(mem_max); mem_min = wave_min_u64(mem_min);
  c_max = wave_max_u64(c_max); c_min = wave_min_u64(c_min);
  qos_max = wave_umax32(qos_max); qos_min = wave_umin32(qos_min);
  part_max = wave_umax32(part_max); part_min = wave_umin32(part_min);
  nn_max = wave_umax32(nn_max); nn_min = wave_umin32(nn_min);
  __shared__ u64 s64[4][6];
  __shared__ u32 s32[4][6];
  const u32 wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63u) == 0) {
    s64[wv][0] = age_max; s64[wv][1] = age_min; s64[wv][2] = mem_max; s64[wv][3] = mem_min; s64[wv][4] = c_max; s64[wv][5] = c_min;
    s32[wv][0] = qos_max; s32[wv][1] = qos_min; s32[wv][2] = part_max; s32[wv][3] = part_min; s32[wv][4] = nn_max; s32[wv][5] = nn_min;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const u32 f = threadIdx.x;
    const bool is_max = (f & 1u) == 0;
    u64 a = s64[0][f];
    u32 b = s32[0][f];
    for (u32 w = 1; w < 4; ++w) {
      a = is_max ? (s64[w][f] > a ? s64[w][f] : a) : (s64[w][f] < a ? s64[w][f] : a);
      b = is_max ? (s32[w][f] > b ? s32[w][f] : b) : (s32[w][f] < b ? s32[w][f] : b);
    }
    PrioBounds* B = P.bounds;
    unsigned long long* p64 = f == 0 ? (unsigned long long*)&B->age_max : f == 1 ? (unsigned long long*)&B->age_min
                            : f == 2 ? (unsigned long long*)&B->mem_max : f == 3 ? (unsigned long long*)&B->mem_min
                            : f == 4 ? (unsigned long long*)&B->cpus_max_bits : (unsigned long long*)&B->cpus_min_bits;
    u32* p32 = f == 0 ? &B->qos_max : f == 1 ? &B->qos_min : f == 2 ? &B->part_max : f == 3 ? &B->part_min
             : f == 4 ? &B->nn_max : &B->nn_min;
    if (is_max) { atomicMax(p64, (unsigned long long)a); atomicMax(p32, b); }
    else { atomicMin(p64, (unsigned long long)a); atomicMin(p32, b); }
  }
}

// Service values (:7715-7746).  The term of a running job (service_val * run_time) does not depend on the
// order, the fp64 SUM of an account does: k_prio_terms computes the terms in parallel, laid out in account-CSR
// order (x-th running job of the account, vector order kept); k_prio_service then adds each account's
// contiguous run front to back, one thread per account (independent loads, only the adds are serial).
__global__ __launch_bounds__(256) void k_prio_terms(const PrioParams P) {
  const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= P.R) return;
  const PrioBounds B = *P.bounds;
  const double cpus_max = __longlong_as_double((long long)B.cpus_max_bits);
  const double cpus_min = __longlong_as_double((long long)B.cpus_min_bits);
  const u32 i = P.acc_jobs[x];
  double service_val = 0;
  if (cpus_max > cpus_min) service_val += 1.0 * (prio_cpu_double(P.r_cpu_raw[i]) - cpus_min) / (cpus_max - cpus_min);
  else service_val += 1.0;
  if (B.nn_max > B.nn_min) service_val += 1.0 * (P.r_node_num[i] - B.nn_min) / (B.nn_max - B.nn_min);
  else service_val += 1.0;
  if (B.mem_max > B.mem_min) service_val += 1.0 * (double)(P.r_mem[i] - B.mem_min) / (double)(B.mem_max - B.mem_min);
  else service_val += 1.0;
  const u64 run_time = (u64)(P.now - P.r_start[i]);
  P.terms[x] = service_val * (double)run_time;
}
__global__ __launch_bounds__(256) void k_prio_service(const PrioParams P) {
  const u32 a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= P.A) return;
  double acc = 0.0;
  const u32 b = P.acc_off[a], e = P.acc_off[a + 1];
  u32 x = b;
  for (; x + 8 <= e; x += 8) {  // eight loads in flight, adds strictly in order
    const double t0 = P.terms[x], t1 = P.terms[x + 1], t2 = P.terms[x + 2], t3 = P.terms[x + 3];
    const double t4 = P.terms[x + 4], t5 = P.terms[x + 5], t6 = P.terms[x + 6], t7 = P.terms[x + 7];
    acc += t0; acc += t1; acc += t2; acc += t3; acc += t4; acc += t5; acc += t6; acc += t7;
  }
  for (; x < e; ++x) acc += P.terms[x];
  P.acc_val[a] = acc;
}
__global__ __launch_bounds__(256) void k_prio_service_bounds(const PrioParams P) {
  __shared__ double s_max[256], s_min[256];
  double mx = 0.0, mn = 4294967295.0;  // :7657-7658 (service_val_min starts at uint32 max)
  for (u32 a = threadIdx.x; a < P.A; a += blockDim.x)
    if (P.acc_present[a]) {
      const double v = P.acc_val[a];
      mx = v > mx ? v : mx;   // std::max(ser_val, max)
      mn = v < mn ? v : mn;   // std::min(ser_val, min)
    }
  s_max[threadIdx.x] = mx; s_min[threadIdx.x] = mn;
  __syncthreads();
  for (u32 w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      s_max[threadIdx.x] = s_max[threadIdx.x + w] > s_max[threadIdx.x] ? s_max[threadIdx.x + w] : s_max[threadIdx.x];
      s_min[threadIdx.x] = s_min[threadIdx.x + w] < s_min[threadIdx.x] ? s_min[threadIdx.x + w] : s_min[threadIdx.x];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { P.bounds->sv_max = s_max[0]; P.bounds->sv_min = s_min[0]; }
}

// k_prio_calc: CalculatePriority_ (:7754-7817) per pending job + the sort key.
__global__ __launch_bounds__(256) void k_prio_calc(const PrioParams P) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.J) return;
  const PrioBounds B = *P.bounds;
  const double cpus_max = __longlong_as_double((long long)B.cpus_max_bits);
  const double cpus_min = __longlong_as_double((long long)B.cpus_min_bits);
  double priority = P.cached ? P.cached[i] : 0.0;
  if (priority == 0.0) {  // :7616
    u64 job_age = (u64)(P.now - P.submit[i]);
    job_age = job_age < P.max_age ? job_age : P.max_age;
    const u32 q = P.qos[i], pp = P.part[i], nn = P.node_num[i];
    const u64 m = P.mem[i];
    const double c = prio_cpu_double(P.cpu_raw[i]);
    double qos_factor = 0, age_factor = 0, partition_factor = 0, job_size_factor = 0, fair_share_factor = 0;
    if (B.age_max > B.age_min) age_factor = 1.0 * (double)(job_age - B.age_min) / (double)(B.age_max - B.age_min);
    if (B.qos_max > B.qos_min) qos_factor = 1.0 * (q - B.qos_min) / (B.qos_max - B.qos_min);
    if (B.part_max > B.part_min) partition_factor = 1.0 * (pp - B.part_min) / (B.part_max - B.part_min);
    if (cpus_max > cpus_min) job_size_factor += 1.0 * (c - cpus_min) / (cpus_max - cpus_min);
    if (B.nn_max > B.nn_min) job_size_factor += 1.0 * (nn - B.nn_min) / (B.nn_max - B.nn_min);
    if (B.mem_max > B.mem_min) job_size_factor += 1.0 * (double)(m - B.mem_min) / (double)(B.mem_max - B.mem_min);
    if (P.favor_small) job_size_factor = 1.0 - job_size_factor / 3;
    else job_size_factor /= 3.0;
    if (B.sv_max > B.sv_min) fair_share_factor = 1.0 - (P.acc_val[P.account[i]] - B.sv_min) / (B.sv_max - B.sv_min);
    priority = P.w_age * age_factor + P.w_part * partition_factor + P.w_size * job_size_factor +
               P.w_fair * fair_share_factor + P.w_qos * qos_factor;
  }
  P.prio[i] = priority;
  // order-preserving u64 image of the double, inverted: ascending keys = descending priority
  const u64 b = (u64)__double_as_longlong(priority);
  const u64 asc = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
  P.keys[i] = ~asc;
}

// ---- stable LSD radix sort, 8-bit digits, (u64 key, u32 value) ------------------------------------------
// Three kernels per pass: per-tile digit histogram, scan of the [digit][tile] table (one block per digit row),
// stable scatter.  A tile = kSortTile consecutive elements handled by one 256-thread block, in chunks of 256 in
// index order; inside a chunk the rank among equal digits comes from wave ballots (match-any over the 8
// digit bits) + a per-wave count table in LDS, so equal keys never change their relative order.
constexpr u32 kSortTile = 4096;

__global__ __launch_bounds__(256) void k_sort_hist(const u64* __restrict__ keys, u32 n, u32 shift, u32* __restrict__ hist,
                                                   u32 ntiles) {
  __shared__ u32 s_h[256];
  s_h[threadIdx.x] = 0;
  __syncthreads();
  const u32 base = blockIdx.x * kSortTile;
  for (u32 c = 0; c < kSortTile; c += 256) {
    const u32 i = base + c + threadIdx.x;
    if (i < n) atomicAdd(&s_h[(u32)(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[threadIdx.x * ntiles + blockIdx.x] = s_h[threadIdx.x];  // digit-major: the scan runs over it linearly
}

// Scan of the [digit][tile] table, two levels: every digit row is scanned by its own block (exclusive, in
// place) and leaves its total in rowtot[digit]; the scatter kernel adds the exclusive prefix over rowtot.
__global__ __launch_bounds__(256) void k_sort_rowscan(u32* __restrict__ hist, u32 ntiles, u32* __restrict__ rowtot) {
  __shared__ u32 s_w[4];
  u32* row = hist + (size_t)blockIdx.x * ntiles;
  const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  u32 carry = 0;
  for (u32 c = 0; c < ntiles; c += 256) {
    const u32 i = c + threadIdx.x;
    const u32 v = i < ntiles ? row[i] : 0u;
    u32 inc = v;
#pragma unroll
    for (u32 off = 1; off < 64; off <<= 1) {
      const u32 t = (u32)__shfl_up((int)inc, (int)off);
      if (lane >= off) inc += t;
    }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    u32 pre = carry;
    for (u32 w = 0; w < wv; ++w) pre += s_w[w];
    if (i < ntiles) row[i] = pre + inc - v;
    carry += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) rowtot[blockIdx.x] = carry;
}

__global__ __launch_bounds__(256) void k_sort_scatter(const u64* __restrict__ kin, const u32* __restrict__ vin,
                                                      u64* __restrict__ kout, u32* __restrict__ vout, u32 n, u32 shift,
                                                      const u32* __restrict__ offs, u32 ntiles,
                                                      const u32* __restrict__ rowtot) {
  __shared__ u32 s_base[256];      // next output slot of each digit for this tile
  __shared__ u32 s_cnt[4][256];    // per wave: elements of the current chunk with that digit
  const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  {  // start of digit d's output range = sum of the totals of the smaller digits (block scan of 256 values)
    const u32 v = rowtot[threadIdx.x];
    u32 inc = v;
#pragma unroll
    for (u32 off = 1; off < 64; off <<= 1) {
      const u32 t = (u32)__shfl_up((int)inc, (int)off);
      if (lane >= off) inc += t;
    }
    if (lane == 63) s_cnt[0][wv] = inc;
    __syncthreads();
    u32 pre = 0;
    for (u32 w = 0; w < wv; ++w) pre += s_cnt[0][w];
    __syncthreads();
    s_base[threadIdx.x] = pre + inc - v + offs[threadIdx.x * ntiles + blockIdx.x];
  }
  const u32 base = blockIdx.x * kSortTile;
  for (u32 c = 0; c < kSortTile; c += 256) {
    for (u32 w = 0; w < 4; ++w) s_cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const u32 i = base + c + threadIdx.x;
    const bool act = i < n;
    u64 k = 0;
    u32 v = 0, d = 0;
    if (act) { k = kin[i]; v = vin[i]; d = (u32)(k >> shift) & 255u; }
    // lanes of this wave holding the same digit
    u64 peers = __ballot(act);
#pragma unroll
    for (u32 b = 0; b < 8; ++b) {
      const u64 m = __ballot(act && ((d >> b) & 1u));
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const u32 rank = (u32)__popcll(peers & ((1ull << lane) - 1ull));
    if (act && rank == 0) s_cnt[wv][d] = (u32)__popcll(peers);
    __syncthreads();
    if (act) {
      u32 o = s_base[d] + rank;
      for (u32 w = 0; w < wv; ++w) o += s_cnt[w][d];
      kout[o] = k;
      vout[o] = v;
    }
    __syncthreads();
    s_base[threadIdx.x] += s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_iota(u32* v, u32 n) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = i;
}

}  // namespace cns

// The host pass of cns_upload_jobs (cranesched_amd/csrc/jobs_host.inc: two passes over chunks of the queue on several host threads)
// against the ONE-thread walk it replaced, written out below as it stood in engine.hip up to round 5 — BasicPriority's truncation
// (JobScheduler.h:185-200), the pre-checks of the ordered loop (JobScheduler.cpp:6744-6761), the split by partition (:6516-6530).
// Random queues (reservations, refused partitions, unknown partitions, skipped jobs, a batch limit, GRES, node lists, shared groups),
// every thread count 1..9, and the two error classes with their precedence.   usage: jobs_host_test <cases> [bench]
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/crane_gpu/node_select.h"
#include "../../cranesched_amd/csrc/jobs_host.inc"

using u64 = uint64_t;
using u32 = uint32_t;
namespace jh = cns_jobs_host;

static u64 g_s = 0x9E3779B97F4A7C15ull;
static u64 rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return g_s; }
static u32 below(u32 n) { return (u32)(rnd() % n); }

struct Queue {
  std::vector<u32> partition, node_num, ntasks, tmin, tmax, reservation;
  std::vector<int64_t> L, tcpu, ncpu;
  std::vector<u64> nmem, tmem, incl_off, excl_off;
  std::vector<uint8_t> excl, gtot, gspec, skip;
  std::vector<u32> incl_nodes, excl_nodes;
  cns_job_soa soa{};
};

struct Snap {
  std::vector<uint8_t> refused, tag;
  std::vector<u32> eng, size, part_off;
  jh::Route R;
};

struct Serial {
  std::vector<uint8_t> reason, jtag;
  std::vector<u32> job_part, grouped;
  std::vector<u64> place_off, pj_off;
  u64 places = 0, algo = 0, n_shaped = 0, Jg = 0;
  int rc = 0;
  std::string err;
};

// the walk as cns_upload_jobs did it on one thread
static void serial(const cns_job_soa* jb, const jh::Route& R, bool shared, Serial& s) {
  const u64 J = jb->num_jobs;
  s.reason.assign(J, CNS_REASON_NONE);
  std::vector<u64> pj_cnt(R.P + 1, 0);
  s.job_part.assign(J, jh::kNoPart);
  std::vector<u32> part_of(J, 0);
  s.place_off.assign(J + 1, 0);
  u64 places = 0;
  for (u64 j = 0; j < J; ++j) {
    s.place_off[j] = places;
    const u32 k = jb->node_num[j];
    if (k == 0 || jb->ntasks[j] < k || jb->ntasks_per_node_min[j] == 0 || jb->ntasks_per_node_max[j] < jb->ntasks_per_node_min[j] ||
        jb->time_limit_sec[j] <= 0 || jb->task_cpu_raw[j] < 0 || (jb->node_cpu_raw && jb->node_cpu_raw[j] < 0)) {
      s.rc = CNS_ERR_INVALID_ARG;
      s.err = "job " + std::to_string(j) + ": invalid node_num/ntasks/time_limit/cpu";
      return;
    }
    places += k;
    if (j >= R.batch) { s.reason[j] = CNS_REASON_PRIORITY; continue; }
    if (jb->skip && jb->skip[j]) { s.reason[j] = CNS_REASON_SKIPPED; continue; }
    const u32 rsv = jb->reservation ? jb->reservation[j] : CNS_RESV_NONE;
    u32 p;
    if (rsv != CNS_RESV_NONE) {
      if (rsv >= R.V) { s.reason[j] = CNS_REASON_RESERVATION_NOT_FOUND; continue; }
      p = R.P_real + rsv;
    } else {
      if (jb->partition[j] >= R.Pu) { s.reason[j] = CNS_REASON_PARTITION_NOT_FOUND; continue; }
      if (R.upart_refused[jb->partition[j]]) { s.reason[j] = CNS_REASON_ENGINE_REFUSED; continue; }
      p = R.upart_eng[jb->partition[j]];
    }
    part_of[j] = p;
    s.job_part[j] = p;
    pj_cnt[p + 1]++;
    bool shaped = k == 1 && jb->ntasks[j] == 1 && jb->ntasks_per_node_min[j] == 1 && !(jb->exclusive && jb->exclusive[j]) &&
                  !(jb->incl_offsets && jb->incl_offsets[j + 1] != jb->incl_offsets[j]) && !(jb->excl_offsets && jb->excl_offsets[j + 1] != jb->excl_offsets[j]);
    if (shaped && jb->gres_total)
      for (u32 x = 0; x < CNS_MAX_GRES_NAMES; ++x) shaped = shaped && jb->gres_total[j * CNS_MAX_GRES_NAMES + x] == 0;
    if (shaped && jb->gres_spec)
      for (u32 x = 0; x < CNS_MAX_GRES_CLASSES; ++x) shaped = shaped && jb->gres_spec[j * CNS_MAX_GRES_CLASSES + x] == 0;
    s.n_shaped += shaped ? 1u : 0u;
    const u64 np = rsv != CNS_RESV_NONE ? (u64)(R.part_off[p + 1] - R.part_off[p]) : (u64)R.upart_size[jb->partition[j]];
    s.algo += np * R.s_node + 64 + 16 + 24ull * k;
  }
  s.place_off[J] = places;
  s.places = places;
  s.pj_off.assign(R.P + 1, 0);
  for (u32 p = 0; p < R.P; ++p) s.pj_off[p + 1] = s.pj_off[p] + pj_cnt[p + 1];
  s.Jg = s.pj_off[R.P];
  std::vector<u64> cur(s.pj_off.begin(), s.pj_off.end() - 1);
  s.grouped.assign(s.Jg, 0);
  for (u64 j = 0; j < R.batch && j < J; ++j) {
    if (s.reason[j] != CNS_REASON_NONE) continue;
    s.grouped[cur[part_of[j]]++] = (u32)j;
  }
  if (jb->gres_spec)
    for (u64 i = 0; i < s.Jg; ++i) {
      const uint8_t* g = jb->gres_spec + (u64)s.grouped[i] * CNS_MAX_GRES_CLASSES;
      for (u32 c = R.gres_classes; c < CNS_MAX_GRES_CLASSES; ++c)
        if (g[c]) { s.rc = CNS_ERR_INVALID_ARG; s.err = "job requests an undefined GRES class"; return; }
    }
  if (shared) {
    s.jtag.assign(J, 0);
    for (u64 j = 0; j < J; ++j)
      if ((!jb->reservation || jb->reservation[j] == CNS_RESV_NONE) && jb->partition[j] < R.Pu) s.jtag[j] = R.upart_tag[jb->partition[j]];
  }
}

static void make_snap(Snap& sn, bool shared) {
  const u32 Pu = 1 + below(12), V = below(4);
  // engine partitions: groups of caller partitions (shared: some share an engine partition)
  sn.eng.assign(Pu, 0); sn.size.assign(Pu, 0); sn.refused.assign(Pu, 0); sn.tag.assign(Pu, 0);
  u32 P_real = 0;
  std::vector<u32> members;
  for (u32 p = 0; p < Pu; ++p) {
    if (shared && P_real && below(3) == 0) { sn.eng[p] = below(P_real); }
    else { sn.eng[p] = P_real++; members.push_back(0); }
    sn.tag[p] = (uint8_t)members[sn.eng[p]]++;
    sn.size[p] = 1 + below(5000);
    if (below(9) == 0) { sn.refused[p] = 1 + below(3); sn.size[p] = 0; }
  }
  const u32 P = P_real + V;
  sn.part_off.assign(P + 1, 0);
  for (u32 p = 0; p < P; ++p) sn.part_off[p + 1] = sn.part_off[p] + 1 + below(3000);
  jh::Route& R = sn.R;
  R.P = P; R.Pu = Pu; R.P_real = P_real; R.V = V;
  R.upart_refused = sn.refused.data(); R.upart_eng = sn.eng.data(); R.upart_size = sn.size.data(); R.upart_tag = sn.tag.data();
  R.part_off = sn.part_off.data();
  R.s_node = below(2) ? 48 : 32;
  R.gres_classes = below(CNS_MAX_GRES_CLASSES + 1);
}

static void make_queue(Queue& q, const Snap& sn, u64 J, int flavour) {
  // flavour 0: valid; 1: a few invalid jobs; 2: an undefined GRES class somewhere; 3: both
  q.partition.resize(J); q.node_num.resize(J); q.ntasks.resize(J); q.tmin.resize(J); q.tmax.resize(J); q.reservation.resize(J);
  q.L.resize(J); q.tcpu.resize(J); q.ncpu.resize(J); q.nmem.resize(J); q.tmem.resize(J);
  q.excl.resize(J); q.gtot.assign(J * CNS_MAX_GRES_NAMES, 0); q.gspec.assign(J * CNS_MAX_GRES_CLASSES, 0); q.skip.resize(J);
  q.incl_off.assign(J + 1, 0); q.excl_off.assign(J + 1, 0); q.incl_nodes.clear(); q.excl_nodes.clear();
  const bool lists = below(2), with_resv = below(2), with_skip = below(2), with_gres = below(4) != 0;
  for (u64 j = 0; j < J; ++j) {
    q.partition[j] = below(10) == 0 ? sn.R.Pu + below(3) : below(sn.R.Pu);
    q.reservation[j] = with_resv && below(6) == 0 ? below(sn.R.V + 2) : CNS_RESV_NONE;
    const u32 k = below(8) == 0 ? 2 + below(7) : 1;
    q.node_num[j] = k;
    q.tmin[j] = 1 + below(2); q.tmax[j] = q.tmin[j] + below(3);
    q.ntasks[j] = below(3) ? k : k * q.tmin[j] + below(4);
    if (q.ntasks[j] < k) q.ntasks[j] = k;
    q.L[j] = 60 * (1 + below(100)); q.tcpu[j] = 256 * (1 + below(8)); q.ncpu[j] = 0;
    q.nmem[j] = rnd() >> 30; q.tmem[j] = rnd() >> 30;
    q.excl[j] = below(20) == 0;
    q.skip[j] = with_skip && below(15) == 0;
    if (with_gres && below(4) == 0) {
      if (below(2)) q.gtot[j * CNS_MAX_GRES_NAMES + below(CNS_MAX_GRES_NAMES)] = 1 + below(8);
      else if (sn.R.gres_classes) q.gspec[j * CNS_MAX_GRES_CLASSES + below(sn.R.gres_classes)] = 1 + below(4);
    }
    if (lists && below(12) == 0) for (u32 i = 0, n = 1 + below(3); i < n; ++i) q.incl_nodes.push_back(below(1000));
    if (lists && below(12) == 0) for (u32 i = 0, n = 1 + below(3); i < n; ++i) q.excl_nodes.push_back(below(1000));
    q.incl_off[j + 1] = q.incl_nodes.size(); q.excl_off[j + 1] = q.excl_nodes.size();
  }
  if (J && (flavour & 1))
    for (u32 i = 0, n = 1 + below(3); i < n; ++i) {
      const u64 j = rnd() % J;
      switch (below(6)) {
        case 0: q.node_num[j] = 0; break;
        case 1: q.ntasks[j] = q.node_num[j] - 1; break;
        case 2: q.tmin[j] = 0; break;
        case 3: q.tmax[j] = q.tmin[j] - 1; break;
        case 4: q.L[j] = 0; break;
        default: q.tcpu[j] = -1; break;
      }
    }
  if (J && (flavour & 2) && sn.R.gres_classes < CNS_MAX_GRES_CLASSES)
    for (u32 i = 0, n = 1 + below(3); i < n; ++i)
      q.gspec[(rnd() % J) * CNS_MAX_GRES_CLASSES + sn.R.gres_classes + below(CNS_MAX_GRES_CLASSES - sn.R.gres_classes)] = 1;
  cns_job_soa& s = q.soa;
  memset(&s, 0, sizeof s);
  s.num_jobs = J;
  s.partition = q.partition.data(); s.time_limit_sec = q.L.data(); s.node_mem = q.nmem.data(); s.task_cpu_raw = q.tcpu.data(); s.task_mem = q.tmem.data();
  s.node_num = q.node_num.data(); s.ntasks = q.ntasks.data(); s.ntasks_per_node_min = q.tmin.data(); s.ntasks_per_node_max = q.tmax.data();
  s.node_cpu_raw = below(2) ? q.ncpu.data() : nullptr;
  s.exclusive = below(3) ? q.excl.data() : nullptr;
  s.gres_total = with_gres || below(2) ? q.gtot.data() : nullptr;
  s.gres_spec = with_gres || (flavour & 2) || below(2) ? q.gspec.data() : nullptr;
  s.skip = with_skip ? q.skip.data() : nullptr;
  s.reservation = with_resv ? q.reservation.data() : nullptr;
  if (lists) { s.incl_offsets = q.incl_off.data(); s.excl_offsets = q.excl_off.data(); s.incl_nodes = q.incl_nodes.data(); s.excl_nodes = q.excl_nodes.data(); }
}

static bool run_parallel(const cns_job_soa* jb, const jh::Route& R, bool shared, u32 threads, Serial& o) {
  const u64 J = jb->num_jobs;
  o = Serial{};
  o.reason.assign(J ? J : 1, 0xEE); o.job_part.assign(J, 0xEEEEEEEEu); o.place_off.assign(J + 1, ~0ull);
  if (shared) o.jtag.assign(J ? J : 1, 0xEE);
  jh::Out O;
  O.reason = o.reason.data(); O.job_part = o.job_part.data(); O.place_off = o.place_off.data(); O.jtag = shared ? o.jtag.data() : nullptr;
  std::vector<jh::Chunk> chunks;
  o.rc = jh::pass1(jb, R, O, chunks, threads, &o.err);
  if (o.rc) return true;
  o.grouped.assign(O.Jg, 0xEEEEEEEEu);
  O.grouped = o.grouped.data();
  jh::pass2(jb, O, chunks);
  o.pj_off = O.pj_off; o.places = O.places; o.algo = O.algo; o.n_shaped = O.n_shaped; o.Jg = O.Jg;
  o.reason.resize(J); if (shared) o.jtag.resize(J);
  return true;
}

#define CHECK(cond, what) do { if (!(cond)) { printf("case %d threads %u: %s differs\n", cs, t, what); return 1; } } while (0)

int main(int argc, char** argv) {
  const int cases = argc > 1 ? atoi(argv[1]) : 300;
  if (argc > 2 && !strcmp(argv[2], "bench")) {   // what the pass costs at BASELINE's queue length on this host
    Snap sn; make_snap(sn, false);
    Queue q; make_queue(q, sn, 1000000, 0);
    sn.R.batch = 1000000;
    const u64 J = 1000000;
    std::vector<uint8_t> reason(J);
    std::vector<u32> job_part(J), grouped(J);
    std::vector<u64> place_off(J + 1);
    for (u32 t : {1u, 2u, 4u, 8u, 16u}) {   // (the engine keeps its staging across cycles: buffers allocated once)
      double best = 1e9, best1 = 1e9;
      for (int rep = 0; rep < 7; ++rep) {
        jh::Out O;
        O.reason = reason.data(); O.job_part = job_part.data(); O.place_off = place_off.data(); O.grouped = grouped.data();
        std::vector<jh::Chunk> chunks;
        std::string err;
        const auto t0 = std::chrono::steady_clock::now();
        if (jh::pass1(&q.soa, sn.R, O, chunks, t, &err)) return 1;
        const auto t1 = std::chrono::steady_clock::now();
        jh::pass2(&q.soa, O, chunks);
        const auto t2 = std::chrono::steady_clock::now();
        const double a = std::chrono::duration<double, std::milli>(t2 - t0).count();
        if (a < best) { best = a; best1 = std::chrono::duration<double, std::milli>(t1 - t0).count(); }
      }
      printf("%2u threads: %.2f ms (pass 1 %.2f, pass 2 %.2f)\n", t, best, best1, best - best1);
    }
    Serial s;
    const auto t0 = std::chrono::steady_clock::now();
    serial(&q.soa, sn.R, false, s);
    printf("one-thread walk of rounds 1-4: %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return 0;
  }
  int n_err = 0, n_ok = 0;
  for (int cs = 0; cs < cases; ++cs) {
    const bool shared = below(3) == 0;
    Snap sn; make_snap(sn, shared);
    const u64 J = cs % 17 == 0 ? below(3) : 1 + below(cs % 5 == 0 ? 40000 : 3000);
    Queue q; make_queue(q, sn, J, cs % 8 < 4 ? cs % 4 : 0);
    sn.R.batch = below(4) == 0 ? (u64)below((u32)J + 1) : J;
    Serial s;
    serial(&q.soa, sn.R, shared, s);
    (s.rc ? n_err : n_ok)++;
    for (u32 t = 1; t <= 9; ++t) {
      Serial o;
      run_parallel(&q.soa, sn.R, shared, t, o);
      CHECK(o.rc == s.rc, "return code");
      CHECK(o.err == s.err, "error text");
      if (s.rc) continue;
      CHECK(o.places == s.places && o.algo == s.algo && o.n_shaped == s.n_shaped && o.Jg == s.Jg, "totals");
      CHECK(o.pj_off == s.pj_off, "pj_off");
      CHECK(o.reason == s.reason, "reason");
      CHECK(o.job_part == s.job_part, "job_part");
      CHECK(o.place_off == s.place_off, "place_off");
      CHECK(o.grouped == s.grouped, "grouped");
      if (shared) CHECK(o.jtag == s.jtag, "jtag");
    }
  }
  printf("ok: %d cases (%d valid queues, %d rejected ones), 1..9 threads each\n", cases, n_ok, n_err);
  return 0;
}

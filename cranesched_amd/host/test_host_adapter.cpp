#include <map>
// Drives GpuNodeSelectionAlgo the way JobScheduler::ScheduleThread_ drives SchedulerAlgo
// (src/CraneCtld/JobScheduler.cpp:1375-1447): build PdJobInScheduler objects, call NodeSelect once,
// read start_time / craned_ids / allocated_res / reason back.  Expected values are the hand-derived
// known answers of tests/kat.py (scenarios A, B and F).
//   test_host_adapter            -> needs an MI355X, exit 0 on success
//   test_host_adapter --no-gpu   -> checks the loud "no device" behaviour instead
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "NodeSelectionAlgo.h"
#include "../../include/crane_gpu/node_select.h"   // cns_group_info (several devices)

using namespace crane;

static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { printf("CHECK failed line %d: %s\n", __LINE__, #c); ++g_fail; } } while (0)

static CranedMeta node(const std::string& id, int cores, uint64_t mem_gib) {
  CranedMeta m;
  m.craned_id = id;
  m.res_total.cpu_set.cpu_count = cpu_t(cores);
  for (int c = 0; c < cores; ++c) m.res_total.cpu_set.core_ids.insert((uint32_t)c);  // CranedMetaContainer.cpp:341-344
  m.res_total.memory_bytes = m.res_total.memory_sw_bytes = mem_gib << 30;
  return m;
}
static std::unique_ptr<PdJobInScheduler> job(job_id_t id, double cpus, int64_t L, const std::string& part = "CPU") {
  auto j = std::make_unique<PdJobInScheduler>();
  j->job_id = id; j->time_limit = L; j->partition_id = part;
  j->req_task_res_view.cpu_count = cpu_t(cpus);
  j->req_task_res_view.memory_bytes = 1ull << 30;
  return j;
}

// Host-side packing of the running jobs with and without the per-job cache; needs no device.
static int pack_bench(int n_nodes, int n_jobs) {
  GpuNodeSelectionAlgo algo(0);   // without a GPU the engine handle is missing; the packing does not need it
  ClusterSnapshot snap;
  std::vector<CranedId> ids;
  for (int i = 0; i < n_nodes; ++i) {
    char name[16];
    snprintf(name, sizeof name, "cn%05d", i);
    snap.craned_metas.push_back(node(name, 64, 256));
    ids.push_back(name);
  }
  snap.partitions = {{"CPU", ids}};
  algo.SetClusterSnapshot(snap);
  std::vector<std::unique_ptr<RnJobInScheduler>> rj;
  uint64_t x = 88172645463325252ull;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  auto mk = [&](job_id_t id) {
    auto r = std::make_unique<RnJobInScheduler>();
    r->job_id = id; r->partition_id = "CPU"; r->start_time = 900; r->end_time = 2000 + (int64_t)(rnd() % 5000);
    const int k = 1 + (int)(rnd() % 4);
    for (int a = 0; a < k; ++a) {
      ResourceInNodeV3& res = r->allocated_res[ids[rnd() % ids.size()]];
      res.cpu_set.cpu_count = cpu_t(8);
      const uint32_t c0 = (uint32_t)(rnd() % 56);
      for (uint32_t c = c0; c < c0 + 8; ++c) res.cpu_set.core_ids.insert(c);
      res.memory_bytes = 16ull << 30;
    }
    return r;
  };
  for (int j = 0; j < n_jobs; ++j) rj.push_back(mk((job_id_t)(j + 1)));
  auto timed = [&](bool cache, uint64_t* sum, size_t* recs) {
    double ms = 0;
    *recs = algo.PackRunningForBench(rj, cache, sum, &ms);
    return ms;
  };
  uint64_t s0, s1, s2, s3, s4;
  size_t n0, n1, n2, n3, n4;
  const double t_full = timed(false, &s0, &n0);     // every cycle from the strings, as the first version did
  const double t_fill = timed(true, &s1, &n1);      // first cycle with the cache: same work + inserts
  const double t_warm = timed(true, &s2, &n2);      // steady state
  // churn: 5 % of the jobs end, as many start
  const int churn = n_jobs / 20;
  rj.erase(rj.begin(), rj.begin() + churn);
  for (int j = 0; j < churn; ++j) rj.push_back(mk((job_id_t)(n_jobs + j + 1)));
  const double t_churn = timed(true, &s3, &n3);
  timed(false, &s4, &n4);
  CHECK(s0 == s1 && s1 == s2 && n0 == n2);
  CHECK(s3 == s4 && n3 == n4);                      // incremental == from scratch after churn
  printf("pack-bench: %d nodes, %d running jobs, %zu allocation records\n", n_nodes, n_jobs, n0);
  printf("  from the strings every cycle : %8.2f ms\n", t_full);
  printf("  first cycle with the cache   : %8.2f ms\n", t_fill);
  printf("  steady state (all cached)    : %8.2f ms  (%.1fx)\n", t_warm, t_full / t_warm);
  printf("  5 %% of the jobs replaced     : %8.2f ms  (%.1fx)\n", t_churn, t_full / t_churn);
  printf("%s\n", g_fail ? "FAIL" : "ok");
  return g_fail != 0;
}

// Event-fed mirror: the adapter is told what CranedMetaContainer is told — MallocResourceFromNode at a job's
// start, FreeResourceFromNode at its end — and must hand the engine the same running tables as the per-cycle walk over
// the running vector (ascending job id).  Host-only.
static int mirror_check(int n_nodes, int n_jobs) {
  GpuNodeSelectionAlgo algo(0), ref(0);
  ClusterSnapshot snap;
  std::vector<CranedId> ids;
  for (int i = 0; i < n_nodes; ++i) {
    char name[16];
    snprintf(name, sizeof name, "cn%05d", i);
    snap.craned_metas.push_back(node(name, 64, 256));
    ids.push_back(name);
  }
  snap.partitions = {{"CPU", ids}};
  algo.SetClusterSnapshot(snap);
  ref.SetClusterSnapshot(snap);
  uint64_t x = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  std::map<job_id_t, std::unique_ptr<RnJobInScheduler>> live;   // what the controller's running-job map holds
  auto start_job = [&](job_id_t id) {
    auto r = std::make_unique<RnJobInScheduler>();
    r->job_id = id; r->partition_id = "CPU"; r->start_time = 900; r->end_time = 2000 + (int64_t)(rnd() % 5000);
    const int k = 1 + (int)(rnd() % 4);
    for (int a = 0; a < k; ++a) {
      ResourceInNodeV3& res = r->allocated_res[ids[rnd() % ids.size()]];
      res.cpu_set.cpu_count = cpu_t(4);
      const uint32_t c0 = (uint32_t)(rnd() % 60);
      res.cpu_set.core_ids.clear();
      for (uint32_t c = c0; c < c0 + 4; ++c) res.cpu_set.core_ids.insert(c);
      res.memory_bytes = 8ull << 30;
    }
    for (const auto& [cid, res] : r->allocated_res) algo.MallocResourceFromNode(cid, id, r->allocated_res);   // cpp:1602-1604
    algo.SetRunningJobInfo(id, r->end_time);
    live[id] = std::move(r);
  };
  auto end_job = [&](job_id_t id) {
    for (const auto& [cid, res] : live[id]->allocated_res) algo.FreeResourceFromNode(cid, id);
    live.erase(id);
  };
  auto same = [&]() {
    std::vector<std::unique_ptr<RnJobInScheduler>> vec;   // the vector NodeSelect would be handed, ascending job id
    for (auto& [id, r] : live) { auto c = std::make_unique<RnJobInScheduler>(*r); vec.push_back(std::move(c)); }
    uint64_t a = 0, b = 0;
    double ms_m = 0, ms_v = 0;
    const size_t na = algo.PackMirrorForBench(&a, &ms_m);
    uint64_t raw;
    const size_t nb = ref.PackRunningForBench(vec, false, &raw, &ms_v);
    b = ref.LastRunningChecksumCanonical();
    CHECK(na == nb && a == b && algo.MirroredRunningJobs() == live.size());
    printf("  %zu running jobs, %zu allocation records: mirror %.2f ms, walk over the running vector %.2f ms, identical: %s\n",
           live.size(), na, ms_m, ms_v, (na == nb && a == b) ? "yes" : "NO");
  };
  job_id_t next = 1;
  for (int j = 0; j < n_jobs; ++j) start_job(next++);
  same();
  for (int round = 0; round < 3; ++round) {   // churn: jobs end in random order, new ones start, one end time changes
    std::vector<job_id_t> idsv;
    for (auto& [id, r] : live) idsv.push_back(id);
    for (int e = 0; e < n_jobs / 10; ++e) {
      const size_t i = rnd() % idsv.size();
      end_job(idsv[i]);
      idsv[i] = idsv.back(); idsv.pop_back();
    }
    for (int e = 0; e < n_jobs / 10; ++e) start_job(next++);
    const job_id_t any = live.begin()->first;
    live[any]->end_time += 777;
    algo.SetRunningJobInfo(any, live[any]->end_time);
    same();
  }
  algo.SetClusterSnapshot(snap);   // a new snapshot re-packs the mirror (dense indices / GRES bits are per snapshot)
  same();
  // ---- the packed form is patched between cycles (ended jobs squeezed out, new ones appended, end times in place); what the patch
  // does not cover must fall back to the full walk — either way the result is the walk's ----------------------------------------
  size_t f0 = 0, p0 = 0, f1 = 0, p1 = 0;
  algo.MirrorPackCounts(&f0, &p0);
  CHECK(p0 >= 3);                                    // the churn rounds above were patches
  same();                                            // nothing happened: an empty patch
  algo.MirrorPackCounts(&f1, &p1);
  CHECK(f1 == f0 && p1 == p0 + 1);
  {   // a running job loses one of its nodes and goes on: not a removal
    job_id_t victim = 0;
    for (auto& [id, r] : live) if (r->allocated_res.size() > 1) { victim = id; break; }
    CHECK(victim != 0);
    const CranedId cid = live[victim]->allocated_res.begin()->first;
    algo.FreeResourceFromNode(cid, victim);
    live[victim]->allocated_res.erase(cid);
    same();
    algo.MirrorPackCounts(&f0, &p0);
    CHECK(f0 == f1 + 1);                             // full walk
  }
  {   // a packed job gets one more allocation
    const job_id_t id = live.begin()->first;
    ResourceV3 one;
    ResourceInNodeV3& res = one[ids[7]];
    res.cpu_set.cpu_count = cpu_t(2); res.cpu_set.core_ids = {60, 61}; res.memory_bytes = 1ull << 30;
    live[id]->allocated_res[ids[7]] = res;
    algo.MallocResourceFromNode(ids[7], id, one);
    same();
  }
  {   // a job id BELOW the packed range starts (ids are not promised to grow), another one ends and comes back within the cycle
    const job_id_t low = live.begin()->first - 1 > 0 && !live.count(live.begin()->first - 1) ? live.begin()->first - 1 : 0;
    if (low) start_job(low);
    const job_id_t again = live.rbegin()->first;
    end_job(again);
    start_job(again);
    same();
  }
  {   // jobs end only; then jobs start only
    std::vector<job_id_t> idsv;
    for (auto& [id, r] : live) idsv.push_back(id);
    for (size_t i = 0; i < idsv.size(); i += 7) end_job(idsv[i]);
    algo.MirrorPackCounts(&f0, &p0);
    same();
    algo.MirrorPackCounts(&f1, &p1);
    CHECK(p1 == p0 + 1 && f1 == f0);                 // removals alone: a patch
    for (int e = 0; e < 1000; ++e) start_job(next++);
    same();
    algo.MirrorPackCounts(&f0, &p0);
    CHECK(p0 == p1 + 1 && f0 == f1);                 // appends alone: a patch
  }
  {   // the first and the last packed job end, every job in between stays
    end_job(live.begin()->first);
    end_job(live.rbegin()->first);
    same();
  }
  {   // a cycle with an explicit running vector in between overwrites the packed arrays
    std::vector<std::unique_ptr<RnJobInScheduler>> vec;
    uint64_t c;
    double ms;
    algo.PackRunningForBench(vec, true, &c, &ms);
    same();
  }
  printf("%s\n", g_fail ? "FAIL" : "ok");
  return g_fail != 0;
}

// Host-side cost of the pending side of one cycle: cns_job_soa packing and the write-back of the placements into the
// PdJobInScheduler objects; needs no device.
static int cycle_bench(int n_nodes, int n_jobs, int threads = 1) {
  GpuNodeSelectionAlgo algo(0);
  algo.SetHostThreads(threads);
  ClusterSnapshot snap;
  std::vector<CranedId> ids;
  for (int i = 0; i < n_nodes; ++i) {
    char name[16];
    snprintf(name, sizeof name, "cn%05d", i);
    snap.craned_metas.push_back(node(name, 64, 256));
    ids.push_back(name);
  }
  snap.partitions = {{"CPU", ids}};
  algo.SetClusterSnapshot(snap);
  std::vector<std::unique_ptr<PdJobInScheduler>> pd;
  for (int j = 0; j < n_jobs; ++j) pd.push_back(job((job_id_t)(j + 1), 4, 600 + j % 1000));
  double pack_ms, wb_ms;
  uint64_t sum;
  algo.PendingCycleForBench(pd, &pack_ms, &wb_ms, &sum);
  CHECK(pd[n_jobs / 2]->allocated_res.begin()->second.cpu_set.core_ids == (std::set<uint32_t>{0, 1, 2, 3}));
  CHECK(pd[n_jobs - 1]->end_time == 1000 + pd[n_jobs - 1]->time_limit && pd[0]->craned_ids[0] == "cn00000");
  printf("cycle-bench: %d nodes, %d pending jobs (all placed, 1 node x 4 cores each), %d host thread%s\n", n_nodes, n_jobs, threads, threads > 1 ? "s" : "");
  printf("  pack  PdJobInScheduler -> cns_job_soa : %8.2f ms = %.2f us / job\n", pack_ms, 1e3 * pack_ms / n_jobs);
  printf("  write placements -> PdJobInScheduler  : %8.2f ms = %.2f us / job\n", wb_ms, 1e3 * wb_ms / n_jobs);
  size_t recs = 0, bytes = 0;
  double wire_ms = algo.EmitWireForBench(&recs, &bytes);
  wire_ms = std::min(wire_ms, algo.EmitWireForBench(&recs, &bytes));
  CHECK(recs == (size_t)n_jobs);
  printf("  placements -> ResourceInNodeV3 wire    : %8.2f ms = %.3f us / job (%zu records, %.1f MB, no objects built)\n", wire_ms,
         1e3 * wire_ms / n_jobs, recs, bytes / 1e6);
  std::string one;
  CHECK(algo.AppendResourceInNodeV3Wire(*pd[n_jobs / 2], pd[n_jobs / 2]->craned_ids[0], &one) && !one.empty());
  {   // deferred write-back + MaterializeAllocation = the full write-back, object for object
    std::vector<std::unique_ptr<PdJobInScheduler>> pd2;
    for (int j = 0; j < n_jobs; ++j) pd2.push_back(job((job_id_t)(j + 1), 4, 600 + j % 1000));
    algo.SetDeferredWriteBack(true);
    double p2, w2;
    uint64_t s2;
    algo.PendingCycleForBench(pd2, &p2, &w2, &s2);
    algo.SetDeferredWriteBack(false);
    printf("  deferred write-back                    : %8.2f ms = %.2f us / job\n", w2, 1e3 * w2 / n_jobs);
    CHECK(pd2[7]->allocated_res.empty() && pd2[7]->craned_ids == pd[7]->craned_ids && pd2[7]->end_time == pd[7]->end_time);
    size_t same = 0;
    for (int j = 0; j < n_jobs; j += 13) {
      CHECK(algo.MaterializeAllocation(*pd2[j]));
      const auto& a = pd[j]->allocated_res.begin()->second;
      const auto& b = pd2[j]->allocated_res.begin()->second;
      same += pd2[j]->allocated_res.size() == 1 && pd[j]->allocated_res.begin()->first == pd2[j]->allocated_res.begin()->first &&
              a.cpu_set.core_ids == b.cpu_set.core_ids && a.memory_bytes == b.memory_bytes && a.memory_sw_bytes == b.memory_sw_bytes &&
              pd2[j]->craned_id_to_task_num == pd[j]->craned_id_to_task_num;
    }
    CHECK(same == (size_t)((n_jobs + 12) / 13));
  }
  printf("%s\n", g_fail ? "FAIL" : "ok");
  return g_fail != 0;
}

// One whole NodeSelect through the adapter at full size, on the GPU: what an integrator's ScheduleThread sees between entering and
// leaving m_node_selection_algo_->NodeSelect (JobScheduler.cpp:1439-1447) — packing, cns_select, write-back — P partitions of N / P nodes
// (64 cores, 256 GiB), J pending jobs of 1..8 cores for 10..170 minutes, spread over the partitions.
// The license pre-pass on a case from a file (tests/test_ref_pin.py writes it and runs the reference's own compiled
// LicenseManager::CheckLicenseCountSufficient on the same arrays): "L J", L lines "total used reserved last_deficit", J lines
// "is_or n (license count) x n" with license indices (>= L: unknown to the table).  Prints per job "rejected n (license count) x n",
// the actual licenses sorted by index.
static int license_file(const char* path) {
  FILE* f = fopen(path, "r");
  if (!f) { printf("cannot open %s\n", path); return 2; }
  unsigned L = 0, J = 0;
  if (fscanf(f, "%u %u", &L, &J) != 2) { fclose(f); return 2; }
  std::unordered_map<std::string, License> table;
  for (unsigned l = 0; l < L; ++l) {
    License lic;
    if (fscanf(f, "%u %u %u %u", &lic.total, &lic.used, &lic.reserved, &lic.last_deficit) != 4) { fclose(f); return 2; }
    table["lic" + std::to_string(l)] = lic;
  }
  std::vector<std::unique_ptr<PdJobInScheduler>> pd;
  std::vector<PdJobInScheduler*> ord;
  for (unsigned j = 0; j < J; ++j) {
    unsigned is_or = 0, n = 0;
    if (fscanf(f, "%u %u", &is_or, &n) != 2) { fclose(f); return 2; }
    auto p = std::make_unique<PdJobInScheduler>();
    p->job_id = j; p->is_license_or = is_or != 0;
    for (unsigned x = 0; x < n; ++x) {
      unsigned lic = 0, cnt = 0;
      if (fscanf(f, "%u %u", &lic, &cnt) != 2) { fclose(f); return 2; }
      p->req_licenses.emplace_back("lic" + std::to_string(lic), cnt);
    }
    p->actual_licenses.emplace("stale", 1u);
    ord.push_back(p.get());
    pd.push_back(std::move(p));
  }
  fclose(f);
  GpuNodeSelectionAlgo::CheckLicenseCountSufficient(table, ord);
  for (const auto& p : pd) {
    std::vector<std::pair<unsigned, unsigned>> a;
    if (!p->req_licenses.empty())
      for (const auto& [id, cnt] : p->actual_licenses) a.emplace_back((unsigned)strtoul(id.c_str() + 3, nullptr, 10), cnt);
    std::sort(a.begin(), a.end());
    printf("%d %zu", p->reason == "License" ? 1 : 0, a.size());
    for (const auto& [l, c] : a) printf(" %u %u", l, c);
    printf("\n");
  }
  return 0;
}

static std::vector<int> parse_devices(const char* s) {   // "0,1,2" (a repeated ordinal — "0,0" — runs several engines on one GPU)
  std::vector<int> d;
  for (const char* p = s; p && *p;) {
    d.push_back(atoi(p));
    p = strchr(p, ',');
    if (p) ++p;
  }
  return d.empty() ? std::vector<int>{0} : d;
}

// Several devices inside the product: ONE GpuNodeSelectionAlgo over `devices` against one over a single device, the same snapshot and queue
// (random partitions, 1..8 cores, one job in seven on two nodes): every job's reason, start, end, nodes, task counts and allocated resources
// must be identical — the shards run on their own host threads, the packed results are all-gathered on the devices and merged in queue order.
static int group_check(int n_nodes, int n_parts, int n_jobs, const std::vector<int>& devices) {
  GpuNodeSelectionAlgo multi(devices), one(devices[0]);
  if (!multi.Ok() || !one.Ok()) { printf("engine: %s%s\n", multi.LastError().c_str(), one.LastError().c_str()); return 2; }
  ClusterSnapshot snap;
  std::vector<std::vector<CranedId>> ids(n_parts);
  for (int i = 0; i < n_nodes; ++i) {
    char name[16];
    snprintf(name, sizeof name, "cn%05d", i);
    snap.craned_metas.push_back(node(name, 64, 256));
    ids[i / ((n_nodes + n_parts - 1) / n_parts)].push_back(name);
  }
  for (int p = 0; p < n_parts; ++p) snap.partitions.push_back({"P" + std::to_string(p), ids[p]});
  multi.SetClusterSnapshot(snap); one.SetClusterSnapshot(snap);
  CHECK(multi.Ok() && one.Ok());
  multi.SetFullWriteBack(true); one.SetFullWriteBack(true);
  std::vector<std::unique_ptr<RnJobInScheduler>> running;
  for (int cycle = 0; cycle < 2; ++cycle) {
    std::vector<std::unique_ptr<PdJobInScheduler>> pa, pb;
    uint64_t x = 0x9E3779B97F4A7C15ull + (uint64_t)cycle;
    for (int j = 0; j < n_jobs; ++j) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      const double cpus = (double)(1 << (x & 3));
      const int64_t L = 600 * (1 + (int)((x >> 8) % 17));
      const std::string part = (j % 97 == 5) ? std::string("nowhere") : "P" + std::to_string((x >> 16) % n_parts);
      pa.push_back(job((job_id_t)(j + 1), cpus, L, part));
      pb.push_back(job((job_id_t)(j + 1), cpus, L, part));
      if (j % 7 == 0) { pa.back()->node_num = pb.back()->node_num = 2; pa.back()->ntasks = pb.back()->ntasks = 2; }
    }
    const auto t0 = std::chrono::steady_clock::now();
    multi.NodeSelect(1000, running, pa);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    one.NodeSelect(1000, running, pb);
    CHECK(multi.Ok() && one.Ok());
    if (!multi.Ok()) printf("multi: %s\n", multi.LastError().c_str());
    size_t same = 0, started = 0;
    for (int j = 0; j < n_jobs; ++j) {
      bool ok = pa[j]->reason == pb[j]->reason && pa[j]->start_time == pb[j]->start_time && pa[j]->end_time == pb[j]->end_time &&
                pa[j]->craned_ids == pb[j]->craned_ids && pa[j]->craned_id_to_task_num == pb[j]->craned_id_to_task_num &&
                pa[j]->allocated_res.size() == pb[j]->allocated_res.size();
      if (ok)
        for (const auto& [cid, rb] : pb[j]->allocated_res) {
          auto it = pa[j]->allocated_res.find(cid);
          ok = ok && it != pa[j]->allocated_res.end() && it->second.cpu_set.cpu_count == rb.cpu_set.cpu_count && it->second.cpu_set.core_ids == rb.cpu_set.core_ids &&
               it->second.memory_bytes == rb.memory_bytes && it->second.gres == rb.gres;
        }
      same += ok;
      started += pb[j]->reason.empty();
    }
    CHECK(same == (size_t)n_jobs);
    cns_group_info gi{};
    const bool have = multi.LastGroupInfo(&gi);
    CHECK(have == (devices.size() > 1));
    printf("  cycle %d: %zu of %d jobs identical on %zu device%s and on one (%zu start now); the multi-device NodeSelect %.1f ms", cycle, same, n_jobs,
           multi.NumDevices(), multi.NumDevices() == 1 ? "" : "s", started, ms);
    if (have) printf(" (shards %.1f | all-gather %.2f [%s, %llu bytes per rank] | download %.2f | scatter %.2f ms; slowest selection kernel %.1f ms)", gi.shards_ms, gi.allgather_ms,
                     gi.gather_mode == CNS_GATHER_RCCL ? "RCCL" : "device-to-device copies", (unsigned long long)gi.slot_bytes, gi.download_ms, gi.scatter_ms, gi.max_select_ms);
    printf("\n");
  }
  printf("%s\n", g_fail ? "FAIL" : "ok");
  return g_fail != 0;
}

static int e2e_bench(int n_nodes, int n_parts, int n_jobs, bool deferred, int threads, const std::vector<int>& devices = {0}) {
  GpuNodeSelectionAlgo algo(devices);
  if (!algo.Ok()) { printf("engine: %s\n", algo.LastError().c_str()); return 2; }
  algo.SetDeferredWriteBack(deferred);
  algo.SetHostThreads(threads);
  ClusterSnapshot snap;
  std::vector<std::vector<CranedId>> ids(n_parts);
  for (int i = 0; i < n_nodes; ++i) {
    char name[16];
    snprintf(name, sizeof name, "cn%05d", i);
    snap.craned_metas.push_back(node(name, 64, 256));
    ids[i / ((n_nodes + n_parts - 1) / n_parts)].push_back(name);
  }
  for (int p = 0; p < n_parts; ++p) snap.partitions.push_back({"P" + std::to_string(p), ids[p]});
  algo.SetClusterSnapshot(snap);
  if (!algo.Ok()) { printf("snapshot: %s\n", algo.LastError().c_str()); return 2; }
  std::vector<std::unique_ptr<RnJobInScheduler>> running;
  printf("e2e-bench: %d nodes in %d partitions, %d pending jobs, %s write-back, one NodeSelect per line (%d host thread%s + %zu engine%s)\n", n_nodes, n_parts, n_jobs,
         deferred ? "deferred (allocated_res on demand)" : "lazy (default)", threads, threads == 1 ? "" : "s", devices.size(), devices.size() == 1 ? "" : "s");
  for (int rep = 0; rep < 4; ++rep) {
    std::vector<std::unique_ptr<PdJobInScheduler>> pd;
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (int j = 0; j < n_jobs; ++j) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      pd.push_back(job((job_id_t)(j + 1), 1 << (x & 3), 600 * (1 + (int)((x >> 8) % 17)), ("P" + std::to_string((x >> 16) % n_parts)).c_str()));
    }
    const auto t0 = std::chrono::steady_clock::now();
    algo.NodeSelect(1000, running, pd);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    CHECK(algo.Ok());
    double a = 0, b = 0, c = 0;
    algo.LastCycleMs(&a, &b, &c);
    size_t now_n = 0, later = 0;
    for (const auto& p : pd) { now_n += p->reason.empty() && p->start_time == 1000; later += !p->reason.empty() && p->start_time > 1000; }
    if (deferred) {   // the jobs a commit loop would launch: every tenth here
      const auto m0 = std::chrono::steady_clock::now();
      size_t n = 0;
      for (size_t j = 0; j < pd.size(); j += 10) n += algo.MaterializeAllocation(*pd[j]);
      const double mm = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - m0).count();
      CHECK(n > 0 && !pd[0]->allocated_res.empty() && pd[1]->allocated_res.empty());
      printf("  (MaterializeAllocation of %zu jobs: %.1f ms = %.2f us each)\n", n, mm, 1e3 * mm / n);
    }
    printf("  cycle %d: %8.1f ms = %.2f M decisions/s  (pack %.1f | cns_select %.1f | write-back %.1f ms; %zu start now, %zu backfilled)%s\n", rep, ms,
           1e-3 * n_jobs / ms, a, b, c, now_n, later, rep == 0 ? "  [first cycle: the page-locked arrays are allocated]" : "");
  }
  if (deferred && n_jobs <= 100000) {   // deferred + MaterializeAllocation against the default write-back of a second adapter, job by job
    GpuNodeSelectionAlgo ref(0);
    ref.SetClusterSnapshot(snap);
    std::vector<std::unique_ptr<PdJobInScheduler>> pa, pb;
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (int j = 0; j < n_jobs; ++j) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      const double cpus = (j % 5 == 0) ? 0.5 + (double)(x & 3) : (double)(1 << (x & 3));   // fractional requests: no core ids
      const int64_t L = 600 * (1 + (int)((x >> 8) % 17));
      const std::string part = "P" + std::to_string((x >> 16) % n_parts);
      pa.push_back(job((job_id_t)(j + 1), cpus, L, part));
      pb.push_back(job((job_id_t)(j + 1), cpus, L, part));
      if (j % 7 == 0) { pa.back()->node_num = pb.back()->node_num = 2; pa.back()->ntasks = pb.back()->ntasks = 2; }
    }
    algo.NodeSelect(1000, running, pa);
    ref.NodeSelect(1000, running, pb);
    CHECK(algo.Ok() && ref.Ok());
    size_t cmp = 0, multi = 0;
    for (int j = 0; j < n_jobs; ++j) {
      CHECK(pa[j]->reason == pb[j]->reason && pa[j]->start_time == pb[j]->start_time && pa[j]->end_time == pb[j]->end_time &&
            pa[j]->craned_ids == pb[j]->craned_ids);
      if (!pb[j]->reason.empty()) { CHECK(pa[j]->allocated_res.empty()); continue; }
      CHECK(pa[j]->allocated_res.empty() && pa[j]->craned_id_to_task_num.empty());   // deferred: nothing built yet
      CHECK(algo.MaterializeAllocation(*pa[j]));
      CHECK(pa[j]->craned_id_to_task_num == pb[j]->craned_id_to_task_num && pa[j]->allocated_res.size() == pb[j]->allocated_res.size());
      for (const auto& [cid, rb] : pb[j]->allocated_res) {
        auto it = pa[j]->allocated_res.find(cid);
        CHECK(it != pa[j]->allocated_res.end());
        if (it == pa[j]->allocated_res.end()) continue;
        const ResourceInNodeV3& ra = it->second;
        CHECK(ra.cpu_set.cpu_count == rb.cpu_set.cpu_count && ra.cpu_set.core_ids == rb.cpu_set.core_ids && ra.memory_bytes == rb.memory_bytes &&
              ra.memory_sw_bytes == rb.memory_sw_bytes && ra.gres == rb.gres);
      }
      ++cmp;
      multi += pb[j]->craned_ids.size() > 1;
    }
    CHECK(cmp > (size_t)n_jobs / 2 && multi > 0);
    printf("  deferred + MaterializeAllocation = the default write-back on %zu started jobs (%zu on two nodes)\n", cmp, multi);
  }
  printf("%s\n", g_fail ? "FAIL" : "ok");
  return g_fail != 0;
}

// Placement -> wire: random packed allocations on a node with two GRES names / three types; for every one
// the adapter's wire bytes and, as text, the fields of the ResourceInNodeV3 object the write-back builds from the same
// packed record.  tests/test_wire.py decodes the bytes with protobuf (message classes built from
// protos/PublicDefs.proto:33-44,63-69,396-409) and compares.  Host-only.
static int wire_dump(const char* path, int n, bool jobtod) {
  GpuNodeSelectionAlgo algo(0);
  ClusterSnapshot snap;
  CranedMeta m = node("cn0", 128, 512);
  for (int i = 0; i < 8; ++i) m.res_total.gres["gpu"]["a100"].insert("/dev/nvidia" + std::to_string(i));
  for (int i = 0; i < 4; ++i) m.res_total.gres["gpu"]["h100"].insert("/dev/nvidia1" + std::to_string(i));
  for (int i = 0; i < 6; ++i) m.res_total.gres["npu"]["910b"].insert("/dev/davinci" + std::to_string(i));
  snap.craned_metas.push_back(m);
  snap.partitions = {{"GPU", {"cn0"}}};
  algo.SetClusterSnapshot(snap);
  FILE* f = fopen(path, "w");
  if (!f) return 1;
  uint64_t x = 0x243F6A8885A308D3ull;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  for (int i = 0; i < n; ++i) {
    const int mode = i % 6;
    uint64_t lo = rnd() & rnd(), hi = rnd() & rnd() & rnd(), g = rnd() & 0x3FFFF;   // 18 slots in the snapshot
    int64_t cpu = 256 * (int64_t)(__builtin_popcountll(lo) + __builtin_popcountll(hi));
    uint64_t mem = (rnd() % 1000) << 20, msw = (rnd() % 3) ? mem : mem + (1ull << 30);
    if (mode == 1) { lo = hi = 0; cpu = 1 + (int64_t)(rnd() % 2000); }      // fractional: no core ids (PublicHeader.h:550-556)
    if (mode == 2) g = 0;
    if (mode == 3) { g &= 0xFF; }                                            // one type only
    if (mode == 4) { lo = hi = 0; cpu = 0; mem = 0; msw = 0; g = 0; }       // all defaults
    uint64_t w2 = 0, w3 = 0;   // core ids 128..255 (two-byte varints on the wire)
    if (mode == 5) {
      hi = ~0ull; lo = ~0ull; w2 = rnd() | rnd(); w3 = (i % 12 == 5) ? ~0ull : rnd() & rnd();
      cpu = 256 * (int64_t)(128 + __builtin_popcountll(w2) + __builtin_popcountll(w3));
    }
    std::string wire;
    ResourceInNodeV3 obj;
    algo.WireOfPackedForTest(cpu, mem, msw, lo, hi, g, &wire, &obj, w2, w3);
    fprintf(f, "REC %zu\nHEX ", wire.size());
    for (unsigned char c : wire) fprintf(f, "%02x", c);
    fprintf(f, "\nCPU %.17g\nMEM %llu %llu\nIDS", (double)obj.cpu_set.cpu_count.raw / 256.0, (unsigned long long)obj.memory_bytes,
            (unsigned long long)obj.memory_sw_bytes);
    for (uint32_t c : obj.cpu_set.core_ids) fprintf(f, " %u", c);
    fprintf(f, "\n");
    for (const auto& [name, tm] : obj.gres)
      for (const auto& [type, slots] : tm) {
        fprintf(f, "GRES %s %s", name.c_str(), type.c_str());
        for (const auto& sl : slots) fprintf(f, " %s", sl.c_str());
        fprintf(f, "\n");
      }
    if (jobtod) {   // crane.grpc.JobToD around the same ResourceInNodeV3 (empty strings / zero ids are not written)
      const uint32_t job_id = (i % 7) ? (uint32_t)(rnd() % 5000000) : 0, uid = (i % 5) ? (uint32_t)(1000 + rnd() % 60000) : 0;
      const std::string part = (i % 4) ? "GPU" : "", acct = "acct" + std::to_string(i % 9), qos = (i % 3) ? "normal" : "",
                        name = (i % 6) ? "job_" + std::to_string(i) : "";
      std::string jw;
      // every third job is an array child: JobToD.array_task (field 16), incl. the all-zero identity (still written: the field is set)
      const GpuNodeSelectionAlgo::ArrayTaskIdentity at{(i % 9) ? (uint32_t)(rnd() % 100000) : 0u, (i % 9) ? (uint32_t)(rnd() % 1000) : 0u};
      const bool is_child = (i % 3) == 0;
      GpuNodeSelectionAlgo::ComposeJobToDWire(job_id, uid, part, acct, qos, name, wire, &jw, is_child ? &at : nullptr);
      fprintf(f, "JOB %u %u %s %s %s %s\n", job_id, uid, part.empty() ? "-" : part.c_str(), acct.c_str(), qos.empty() ? "-" : qos.c_str(), name.empty() ? "-" : name.c_str());
      if (is_child) fprintf(f, "ARRAY %u %u\n", at.array_job_id, at.task_id);
      fprintf(f, "JOBHEX ");
      for (unsigned char c : jw) fprintf(f, "%02x", c);
      fprintf(f, "\n");
    }
    fprintf(f, "END\n");
  }
  fclose(f);
  printf("wire-dump: %d records -> %s\n", n, path);
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "--cycle-bench")) return cycle_bench(argc > 2 ? atoi(argv[2]) : 16384, argc > 3 ? atoi(argv[3]) : 200000, argc > 4 ? atoi(argv[4]) : 1);
  if (argc > 1 && !strcmp(argv[1], "--e2e-bench")) {   // ... [--devices 0,0] at the end: several engines (cns_group)
    std::vector<int> dev{0};
    for (int a = 2; a + 1 < argc; ++a) if (!strcmp(argv[a], "--devices")) { dev = parse_devices(argv[a + 1]); argc = a; break; }
    return e2e_bench(argc > 2 ? atoi(argv[2]) : 65536, argc > 3 ? atoi(argv[3]) : 8, argc > 4 ? atoi(argv[4]) : 1000000, argc > 5 && !strcmp(argv[5], "deferred"), argc > 6 ? atoi(argv[6]) : 1, dev);
  }
  if (argc > 2 && !strcmp(argv[1], "--license-file")) return license_file(argv[2]);
  if (argc > 1 && !strcmp(argv[1], "--refusal-check")) {
    // one node of partition P2 has a core id the engine cannot carry: P2's jobs come back "GpuEngineRefused" (RefusedJobs lists exactly them),
    // the jobs of P0, P1, P3 are placed exactly as by an adapter whose snapshot has no such node (the partitions are independent)
    GpuNodeSelectionAlgo algo(0), clean(0);
    if (!algo.Ok()) { printf("engine: %s\n", algo.LastError().c_str()); return 2; }
    ClusterSnapshot snap;
    std::vector<std::vector<CranedId>> ids(4);
    for (int i = 0; i < 64; ++i) {
      char name[16];
      snprintf(name, sizeof name, "cn%03d", i);
      snap.craned_metas.push_back(node(name, 8, 32));
      ids[i / 16].push_back(name);
    }
    for (int p = 0; p < 4; ++p) snap.partitions.push_back({"P" + std::to_string(p), ids[p]});
    ClusterSnapshot snap_clean = snap;
    snap.craned_metas[37].res_total.cpu_set.core_ids.insert(511);   // a 512-core machine in P2
    algo.SetClusterSnapshot(snap); clean.SetClusterSnapshot(snap_clean);
    CHECK(algo.Ok() && clean.Ok() && algo.UnsupportedNodes() == 1);
    const auto rp = algo.RefusedPartitions();
    CHECK(rp.size() == 1 && rp[0] == "P2" && clean.RefusedPartitions().empty());
    std::vector<std::unique_ptr<RnJobInScheduler>> running;
    std::vector<std::unique_ptr<PdJobInScheduler>> pa, pb;
    for (int j = 0; j < 400; ++j) {
      const std::string part = "P" + std::to_string((j * 7 + j / 5) % 4);
      pa.push_back(job((job_id_t)(j + 1), 1 + j % 4, 600 * (1 + j % 5), part));
      pb.push_back(job((job_id_t)(j + 1), 1 + j % 4, 600 * (1 + j % 5), part));
    }
    algo.SetFullWriteBack(true); clean.SetFullWriteBack(true);
    algo.NodeSelect(1000, running, pa); clean.NodeSelect(1000, running, pb);
    CHECK(algo.Ok() && clean.Ok());
    size_t refused = 0, same = 0, other = 0;
    for (int j = 0; j < 400; ++j) {
      if (pa[j]->partition_id == "P2") {
        refused += pa[j]->reason == "GpuEngineRefused" && pa[j]->craned_ids.empty() && pa[j]->start_time == 0;
      } else {
        ++other;
        same += pa[j]->reason == pb[j]->reason && pa[j]->start_time == pb[j]->start_time && pa[j]->craned_ids == pb[j]->craned_ids &&
                pa[j]->allocated_res.size() == pb[j]->allocated_res.size();
      }
    }
    CHECK(refused == 400 - other && refused > 50 && same == other);
    CHECK(algo.RefusedJobs().size() == refused && clean.RefusedJobs().empty());
    for (const PdJobInScheduler* p : algo.RefusedJobs()) CHECK(p->partition_id == "P2");
    printf("  %zu jobs of P2 refused (a 512-core node), %zu jobs of the other partitions placed as without it\n", refused, same);
    printf("%s\n", g_fail ? "FAIL" : "ok");
    return g_fail != 0;
  }
  if (argc > 1 && !strcmp(argv[1], "--group-check"))
    return group_check(argc > 2 ? atoi(argv[2]) : 8192, argc > 3 ? atoi(argv[3]) : 8, argc > 4 ? atoi(argv[4]) : 60000, parse_devices(argc > 5 ? argv[5] : "0,0"));
  if (argc > 1 && !strcmp(argv[1], "--pack-bench")) return pack_bench(argc > 2 ? atoi(argv[2]) : 16384, argc > 3 ? atoi(argv[3]) : 100000);
  if (argc > 2 && !strcmp(argv[1], "--wire-dump")) return wire_dump(argv[2], argc > 3 ? atoi(argv[3]) : 600, argc > 4 && !strcmp(argv[4], "jobtod"));
  if (argc > 1 && !strcmp(argv[1], "--mirror-check")) return mirror_check(argc > 2 ? atoi(argv[2]) : 4096, argc > 3 ? atoi(argv[3]) : 20000);
  if (argc > 1 && !strcmp(argv[1], "--config-checks")) {
    // configurations the engine cannot serve are refused when the SNAPSHOT is set (host-only; no device needed), so
    // that the integrator keeps the CPU SchedulerAlgo instead of seeing "GpuEngineError" on every job of every cycle
    GpuNodeSelectionAlgo algo(0);
    ClusterSnapshot snap;
    snap.craned_metas = {node("cn0", 4, 16), node("cn1", 4, 16)};
    snap.partitions = {{"CPU", {"cn0", "cn1"}}};
    snap.preempt_enabled = true;           // preemption together with partitions that share a node is served
    snap.partitions.push_back({"ALL", {"cn0", "cn1"}});
    algo.SetClusterSnapshot(snap);
    CHECK(algo.LastStatus() != -4 /* CNS_ERR_UNSUPPORTED */ && algo.LastError().find("preemption") == std::string::npos);
    snap.preempt_enabled = false;
    snap.partitions.pop_back();
    snap.craned_metas[1].res_total.cpu_set.core_ids.insert(130);   // a 192-core node: ids 128..255 are carried (ABI 3)
    algo.SetClusterSnapshot(snap);
    CHECK(algo.LastStatus() != -4 && algo.LastError().find("core id") == std::string::npos);
    snap.craned_metas[1].res_total.cpu_set.core_ids.insert(300);   // beyond the four mask words: not dropped — the NODE is flagged, the engine
    algo.SetClusterSnapshot(snap);                                  // then refuses the partitions that list it and serves the others (round 5)
    CHECK(algo.UnsupportedNodes() == 1 && algo.LastError().find("core id") == std::string::npos);
    // ... and a RUNNING job that holds such an id refuses every cycle it is part of, not only the one that first packed it: the bit
    // lives with the cached / mirrored allocation record (the flag used to be cleared after one refused cycle, and the
    // next cycle served the cached record with the id silently dropped)
    snap.craned_metas[1].res_total.cpu_set.core_ids.erase(300);
    algo.SetClusterSnapshot(snap);
    {
      std::vector<std::unique_ptr<RnJobInScheduler>> rn;
      auto r1 = std::make_unique<RnJobInScheduler>();
      r1->job_id = 7; r1->end_time = 5000;
      r1->allocated_res["cn1"].cpu_set.cpu_count = cpu_t::from_raw(2 * 256);
      r1->allocated_res["cn1"].cpu_set.core_ids = {1, 300};
      auto r2 = std::make_unique<RnJobInScheduler>();
      r2->job_id = 8; r2->end_time = 5000;
      r2->allocated_res["cn0"].cpu_set.cpu_count = cpu_t::from_raw(256);
      r2->allocated_res["cn0"].cpu_set.core_ids = {0};
      rn.push_back(std::move(r1)); rn.push_back(std::move(r2));
      algo.PackRunningForBench(rn, true, nullptr, nullptr);
      CHECK(algo.PackedRunningSetOverflows());
      algo.PackRunningForBench(rn, true, nullptr, nullptr);      // the second cycle: served from the per-job cache
      CHECK(algo.PackedRunningSetOverflows());
      rn.erase(rn.begin());                                      // the job ended
      algo.PackRunningForBench(rn, true, nullptr, nullptr);
      CHECK(!algo.PackedRunningSetOverflows());
      ResourceV3 res;
      res["cn1"].cpu_set.cpu_count = cpu_t::from_raw(256);
      res["cn1"].cpu_set.core_ids = {299};
      algo.MallocResourceFromNode("cn1", 9, res);
      algo.SetRunningJobInfo(9, 6000, "");
      algo.PackMirrorForBench(nullptr);
      CHECK(algo.PackedRunningSetOverflows());
      algo.PackMirrorForBench(nullptr);                          // patched pack: nothing changed
      CHECK(algo.PackedRunningSetOverflows());
      algo.FreeResourceFromNode("cn1", 9);
      algo.PackMirrorForBench(nullptr);
      CHECK(!algo.PackedRunningSetOverflows());
    }
    printf("%s\n", g_fail ? "FAIL" : "ok");
    return g_fail != 0;
  }
  const bool no_gpu = argc > 1 && !strcmp(argv[1], "--no-gpu");
  const TimeSec now = 1000;
  std::vector<std::unique_ptr<RnJobInScheduler>> running;
  {
    GpuNodeSelectionAlgo algo(0);
    if (no_gpu) {
      // no device: construction reports it, NodeSelect leaves every job unscheduled with a reason — no CPU path
      std::vector<std::unique_ptr<PdJobInScheduler>> pd;
      pd.push_back(job(1, 1, 100));
      algo.NodeSelect(now, running, pd);
      if (algo.Ok()) { printf("a GPU is present; nothing to check in --no-gpu mode\n"); return 0; }
      CHECK(algo.LastStatus() == -2);
      CHECK(pd[0]->reason == "GpuEngineError" && !pd[0]->is_scheduled());
      printf("%s (%s)\n", g_fail ? "FAIL" : "ok", algo.LastError().c_str());
      return g_fail != 0;
    }
    if (!algo.Ok()) { printf("engine: %s\n", algo.LastError().c_str()); return 2; }

    // --- scenario A: min-load-first with index tie-break ------------------------------------------------
    ClusterSnapshot snap;
    snap.craned_metas = {node("cn0", 4, 16), node("cn1", 4, 16), node("cn2", 4, 16)};
    snap.partitions = {{"CPU", {"cn0", "cn1", "cn2"}}};
    algo.SetClusterSnapshot(snap);
    std::vector<std::unique_ptr<PdJobInScheduler>> pd;
    pd.push_back(job(1, 1, 100)); pd.push_back(job(2, 1, 100)); pd.push_back(job(3, 1, 100)); pd.push_back(job(4, 1, 50));
    pd.push_back(job(5, 1, 10, "NOPE"));
    algo.NodeSelect(now, running, pd);
    CHECK(algo.Ok());
    const char* want[] = {"cn0", "cn1", "cn2", "cn0"};
    for (int i = 0; i < 4; ++i) {
      CHECK(pd[i]->is_scheduled() && pd[i]->start_time == now && pd[i]->end_time == now + pd[i]->time_limit);
      CHECK(pd[i]->craned_ids.size() == 1 && pd[i]->craned_ids[0] == want[i]);
      CHECK(pd[i]->craned_id_to_task_num.at(want[i]) == 1);
      CHECK(pd[i]->allocated_res.at(want[i]).cpu_set.cpu_count == cpu_t(1));
    }
    CHECK(pd[3]->allocated_res.at("cn0").cpu_set.core_ids == std::set<uint32_t>{1});  // lowest free core id
    CHECK(pd[4]->reason == "Partition Not Found");

    // --- a cycle the engine refuses must not leave the PREVIOUS cycle's placements behind for MaterializeAllocation,
    // and a running job with a core id >= 256 refuses every cycle it is part of (the second one is served from the per-job cache)
    {
      algo.SetDeferredWriteBack(true);
      std::vector<std::unique_ptr<PdJobInScheduler>> pq;
      pq.push_back(job(1, 1, 100));
      algo.NodeSelect(now, running, pq);
      CHECK(algo.Ok() && pq[0]->is_scheduled() && algo.MaterializeAllocation(*pq[0]));
      std::vector<std::unique_ptr<RnJobInScheduler>> bad;
      auto r1 = std::make_unique<RnJobInScheduler>();
      r1->job_id = 77; r1->end_time = now + 500;
      r1->allocated_res["cn1"].cpu_set.cpu_count = cpu_t::from_raw(256);
      r1->allocated_res["cn1"].cpu_set.core_ids = {300};
      bad.push_back(std::move(r1));
      for (int rep = 0; rep < 2; ++rep) {
        pq[0]->reason.clear();
        algo.NodeSelect(now, bad, pq);
        CHECK(!algo.Ok() && algo.LastStatus() == -4 && algo.LastError().find("core id") != std::string::npos);
        CHECK(pq[0]->reason == "GpuEngineError");
        CHECK(!algo.MaterializeAllocation(*pq[0]));
      }
      pq[0]->reason.clear();
      algo.NodeSelect(now, running, pq);   // the job ended: the next cycle is served again
      CHECK(algo.Ok() && pq[0]->is_scheduled() && algo.MaterializeAllocation(*pq[0]));
      algo.SetDeferredWriteBack(false);
    }

    // --- the same queue after cn0 went down (CranedDown) and came back: SetCranedState re-sends the packed tables,
    // no new snapshot; a node that is not alive is skipped (JobScheduler.cpp:6595) -------------------------------
    {
      auto again = [&]() {
        pd.clear();
        pd.push_back(job(1, 1, 100)); pd.push_back(job(2, 1, 100)); pd.push_back(job(3, 1, 100));
        algo.NodeSelect(now, running, pd);
      };
      algo.SetCranedState("cn0", /*alive=*/false, /*drain=*/false);
      CHECK(algo.Ok());
      again();
      CHECK(pd[0]->craned_ids[0] == "cn1" && pd[1]->craned_ids[0] == "cn2" && pd[2]->craned_ids[0] == "cn1");
      algo.SetCranedState("cn0", true, true);   // alive but draining: still skipped
      again();
      CHECK(pd[0]->craned_ids[0] == "cn1");
      algo.SetCranedState("cn0", true, false);
      again();
      CHECK(pd[0]->craned_ids[0] == "cn0" && pd[1]->craned_ids[0] == "cn1" && pd[2]->craned_ids[0] == "cn2");
      algo.SetCranedState("cn9", true, false);  // unknown craned: reported, nothing changes
      CHECK(!algo.Ok());
      again();
      CHECK(algo.Ok() && pd[0]->craned_ids[0] == "cn0");
    }

    // --- scenario B: backfill, reasons ---------------------------------------------------------------------
    snap.craned_metas = {node("cn0", 2, 8)};
    snap.partitions = {{"CPU", {"cn0"}}};
    algo.SetClusterSnapshot(snap);
    pd.clear();
    pd.push_back(job(1, 2, 100)); pd.push_back(job(2, 1, 50)); pd.push_back(job(3, 2, 10)); pd.push_back(job(4, 1, 10));
    algo.NodeSelect(now, running, pd);
    CHECK(pd[0]->is_scheduled() && pd[0]->start_time == now);
    CHECK(pd[1]->reason == "Priority" && pd[1]->start_time == 1100);
    CHECK(pd[2]->reason == "Priority" && pd[2]->start_time == 1150);
    CHECK(pd[3]->reason == "Priority" && pd[3]->start_time == 1100);
    // default (lazy) write-back: a job that did not start now carries reason + start time only (cpp:1503-1510 reads no more)
    CHECK(pd[1]->craned_ids.empty() && pd[1]->allocated_res.empty());
    algo.SetFullWriteBack(true);   // ... the reference's own NodeSelect also leaves its backfill placement behind
    pd.clear();
    pd.push_back(job(1, 2, 100)); pd.push_back(job(2, 1, 50));
    algo.NodeSelect(now, running, pd);
    CHECK(pd[1]->reason == "Priority" && pd[1]->start_time == 1100 && pd[1]->craned_ids.size() == 1 && pd[1]->craned_ids[0] == "cn0");
    CHECK(pd[1]->allocated_res.at("cn0").cpu_set.core_ids == std::set<uint32_t>{0});   // allocated against res_total (:6354-6356)
    algo.SetFullWriteBack(false);

    // --- scenario F: GRES slot choice, slot paths in lexicographic order -------------------------------------
    CranedMeta g0 = node("gn0", 8, 16), g1 = node("gn1", 8, 16);
    for (auto* g : {&g0, &g1}) {
      for (int i = 0; i < 4; ++i) g->res_total.gres["gpu"]["a100"].insert("/dev/nvidia" + std::to_string(i));
      for (int i = 4; i < 8; ++i) g->res_total.gres["gpu"]["h100"].insert("/dev/nvidia" + std::to_string(i));
    }
    snap.craned_metas = {g0, g1};
    snap.partitions = {{"GPU", {"gn0", "gn1"}}};
    algo.SetClusterSnapshot(snap);
    pd.clear();
    auto a = job(1, 1, 100, "GPU");
    a->req_node_res_view.gres_map["gpu"].total = 3;
    a->req_node_res_view.gres_map["gpu"].specified["h100"] = 1;
    auto b = job(2, 1, 100, "GPU");
    b->req_node_res_view.gres_map["gpu"].total = 2;
    pd.push_back(std::move(a)); pd.push_back(std::move(b));
    algo.NodeSelect(now, running, pd);
    CHECK(pd[0]->is_scheduled() && pd[0]->craned_ids[0] == "gn0");
    CHECK((pd[0]->allocated_res.at("gn0").gres.at("gpu").at("h100") ==
           std::set<SlotId>{"/dev/nvidia4", "/dev/nvidia5", "/dev/nvidia6"}));
    CHECK(pd[1]->is_scheduled() && pd[1]->craned_ids[0] == "gn1");
    CHECK((pd[1]->allocated_res.at("gn1").gres.at("gpu").at("a100") == std::set<SlotId>{"/dev/nvidia0", "/dev/nvidia1"}));

    // --- license pre-pass (LicenseManager.cpp:167-221) ----------------------------------------------------
    {
      snap.craned_metas = {node("cn0", 8, 64)};
      snap.partitions = {{"CPU", {"cn0"}}};
      algo.SetClusterSnapshot(snap);
      algo.SetLicenses({{"matlab", License{3, 1, 0, 0}}, {"ansys", License{1, 0, 1, 0}}});
      pd.clear();
      auto a = job(1, 1, 10); a->req_licenses = {{"matlab", 2}};                      // 2 + 1 used <= 3: ok, used -> 3
      auto b = job(2, 1, 10); b->req_licenses = {{"matlab", 1}};                      // 1 + 3 > 3: "License"
      auto c = job(3, 1, 10); c->req_licenses = {{"ansys", 1}, {"matlab", 1}}; c->is_license_or = true;  // neither fits
      auto d = job(4, 1, 10); d->req_licenses = {{"nope", 1}};                        // unknown license
      auto e = job(5, 1, 10);                                                         // no license request
      pd.push_back(std::move(a)); pd.push_back(std::move(b)); pd.push_back(std::move(c)); pd.push_back(std::move(d));
      pd.push_back(std::move(e));
      std::vector<std::unique_ptr<RnJobInScheduler>> none;
      algo.NodeSelect(now, none, pd);
      CHECK(algo.Ok());
      CHECK(pd[0]->is_scheduled() && pd[0]->actual_licenses.at("matlab") == 2);
      CHECK(pd[1]->reason == "License" && pd[2]->reason == "License" && pd[3]->reason == "License");
      CHECK(pd[4]->is_scheduled());
      CHECK(pd[4]->allocated_res.at("cn0").cpu_set.core_ids == (std::set<uint32_t>{1}));  // rejected jobs took nothing
      algo.SetLicenses({});
    }

    // --- reservations (JobScheduler.cpp:6619-6679, 6754-6760, 6797-6806) -----------------------------------
    {
      snap.craned_metas = {node("cn0", 8, 16)};
      snap.partitions = {{"CPU", {"cn0"}}};
      ResvMeta rv0; rv0.name = "r0"; rv0.start_time = now - 10; rv0.end_time = now + 500;
      rv0.res_total["cn0"].cpu_set.cpu_count = cpu_t(4); rv0.res_total["cn0"].cpu_set.core_ids = {4, 5, 6, 7};
      rv0.res_total["cn0"].memory_bytes = 8ull << 30;
      ResvMeta rv1; rv1.name = "later"; rv1.start_time = now + 1000; rv1.end_time = now + 2000;
      rv1.res_total["cn0"].cpu_set.cpu_count = cpu_t(1); rv1.res_total["cn0"].cpu_set.core_ids = {0};
      rv1.res_total["cn0"].memory_bytes = 1ull << 30;
      snap.reservations = {rv0, rv1};
      algo.SetClusterSnapshot(snap);
      CHECK(algo.Ok());
      pd.clear();
      pd.push_back(job(1, 6, 100));                                   // 6 cpus: only 4 outside the reservation
      auto in = job(2, 2, 100); in->reservation = "r0"; pd.push_back(std::move(in));
      auto fut = job(3, 1, 10); fut->reservation = "later"; pd.push_back(std::move(fut));
      auto unk = job(4, 1, 10); unk->reservation = "nope"; pd.push_back(std::move(unk));
      std::vector<std::unique_ptr<RnJobInScheduler>> none;
      algo.NodeSelect(now, none, pd);
      CHECK(algo.Ok());
      CHECK(pd[0]->reason == "Resource Reserved" && pd[0]->start_time == now + 500);
      CHECK(pd[1]->is_scheduled() && pd[1]->start_time == now);
      CHECK(pd[1]->allocated_res.at("cn0").cpu_set.core_ids == (std::set<uint32_t>{4, 5}));  // inside the reserved cores
      CHECK(pd[2]->reason == "Reservation Not Found" && pd[3]->reason == "Reservation Not Found");
      snap.reservations.clear();
    }

    // --- MultiFactorPriority in front of the selection (JobScheduler.cpp:6735) ------------------------------
    {
      PriorityConfig pc;
      pc.MaxAge = 500; pc.WeightAge = 1000; pc.WeightFairShare = 2000; pc.WeightJobSize = 300; pc.WeightPartition = 40; pc.WeightQoS = 5;
      GpuMultiFactorPriority sorter(pc, 0);
      CHECK(sorter.Ok());
      snap.craned_metas = {node("cn0", 8, 64)};
      snap.partitions = {{"CPU", {"cn0"}}};
      algo.SetClusterSnapshot(snap);
      algo.SetPrioritySorter(&sorter);
      pd.clear();
      // same numbers as tests/test_priority.py::_kat -> expected order job 3, job 1, job 2
      auto mk = [&](job_id_t id, int64_t age, uint32_t qos, uint32_t part, uint32_t nodes, int cpus, uint64_t gib, const char* acc) {
        auto j = job(id, cpus, 100);
        j->submit_time = now - age; j->qos_priority = qos; j->partition_priority = part; j->account = acc;
        j->req_total_res_view.cpu_count = cpu_t(cpus); j->req_total_res_view.memory_bytes = gib << 30;
        j->node_num = 1; (void)nodes;
        return j;
      };
      pd.push_back(mk(1, 100, 0, 1, 1, 1, 1, "a0"));
      pd.push_back(mk(2, 300, 10, 1, 1, 4, 4, "a1"));
      pd.push_back(mk(3, 9999, 10, 5, 1, 2, 2, "a0"));
      std::vector<std::unique_ptr<RnJobInScheduler>> rj;
      auto r0 = std::make_unique<RnJobInScheduler>();
      r0->start_time = now - 1000; r0->end_time = now + 5; r0->account = "a0"; r0->node_num = 1;
      r0->allocated_res_view.cpu_count = cpu_t(2); r0->allocated_res_view.memory_bytes = 2ull << 30; r0->partition_priority = 1;
      auto r1 = std::make_unique<RnJobInScheduler>();
      r1->start_time = now - 2000; r1->end_time = now + 5; r1->account = "a1"; r1->node_num = 4; r1->qos_priority = 10;
      r1->allocated_res_view.cpu_count = cpu_t(8); r1->allocated_res_view.memory_bytes = 8ull << 30; r1->partition_priority = 5;
      rj.push_back(std::move(r0)); rj.push_back(std::move(r1));
      std::vector<PdJobInScheduler*> ordered;
      sorter.GetOrderedJobPtrVec(now, pd, rj, 2, ordered);
      CHECK(sorter.Ok());
      CHECK(ordered.size() == 2 && ordered[0]->job_id == 3 && ordered[1]->job_id == 1);
      CHECK(pd[1]->reason == "Priority");                      // past the limit (cpp:7625-7630)
      CHECK(pd[2]->priority > pd[0]->priority && pd[0]->priority > pd[1]->priority);
      pd[1]->reason.clear();
      algo.NodeSelect(now, running, pd);                       // the 8-cpu node takes them in priority order
      CHECK(pd[2]->is_scheduled() && pd[0]->is_scheduled() && pd[1]->is_scheduled());
      CHECK(pd[2]->allocated_res.at("cn0").cpu_set.core_ids == (std::set<uint32_t>{0, 1}));   // first served: lowest cores
      CHECK(pd[0]->allocated_res.at("cn0").cpu_set.core_ids == (std::set<uint32_t>{2}));
      algo.SetPrioritySorter(nullptr);
    }

    // --- the commit loop's run-limit admission after NodeSelect (JobScheduler.cpp:1557-1573) --------------------
    {
      snap.craned_metas = {node("cn0", 4, 16), node("cn1", 4, 16)};
      snap.partitions = {{"CPU", {"cn0", "cn1"}}};
      algo.SetClusterSnapshot(snap);
      AccountMetaSnapshot meta;
      Qos normal;
      normal.max_jobs_per_user = 1;                       // -> "QosJobsResourceLimit" for a user's second job (:526-527)
      normal.max_tres_per_account.cpu_count = cpu_t(2);   // every account of the chain may hold 2 cores (:538)
      meta.qos["normal"] = normal;
      meta.account_parent = {{"root", ""}, {"lab", "root"}};
      for (const char* u : {"alice", "bob", "dave"}) {
        meta.user_accounts[u]["lab"];                      // User::account_to_attrs_map, no partition limits
        meta.user_meta[u].qos_to_resource_map["normal"];   // entries created at submit time
      }
      for (const char* a : {"root", "lab"}) meta.account_meta[a].qos_to_resource_map["normal"];
      meta.qos_meta["normal"];
      pd.clear();
      auto uj = [&](job_id_t id, const char* user) {
        auto j = job(id, 1, 100);
        j->username = user; j->account = "lab"; j->qos = "normal";
        return j;
      };
      pd.push_back(uj(1, "alice")); pd.push_back(uj(2, "alice")); pd.push_back(uj(3, "bob")); pd.push_back(uj(4, "carol"));
      pd.push_back(uj(5, "dave"));
      std::vector<std::unique_ptr<RnJobInScheduler>> none;
      algo.NodeSelect(now, none, pd);
      for (const auto& j : pd) CHECK(j->is_scheduled() && j->start_time == now);
      std::vector<std::string> res;
      algo.CheckAndMallocMetaResource(meta, pd, res);
      CHECK(algo.Ok());
      CHECK(res.size() == 5 && res[0].empty() && res[1] == "QosJobsResourceLimit" && res[2].empty());
      CHECK(res[3] == "InvalidUser");                      // not in AccountManager (:186-191)
      CHECK(res[4] == "QosCpuResourceLimit");                 // lab already holds 2 cores (alice, bob)
      CHECK(meta.user_meta["alice"].qos_to_resource_map["normal"].jobs_count == 1);
      CHECK(meta.user_meta["alice"].account_to_partition_to_resource_map["lab"]["CPU"].jobs_count == 1);  // created by DoMallocResource_
      CHECK(meta.account_meta["root"].qos_to_resource_map["normal"].jobs_count == 2);
      CHECK(meta.account_meta["lab"].qos_to_resource_map["normal"].resource.cpu_count == cpu_t(2));
      CHECK(meta.account_meta["lab"].partition_to_resource_map["CPU"].wall_time == 200);
      CHECK(meta.qos_meta["normal"].jobs_count == 2 && meta.user_meta["dave"].qos_to_resource_map["normal"].jobs_count == 0);
    }

    // --- step scheduling inside two jobs' allocations (JobScheduler.cpp:1992-2001; tests/test_steps.py "fifo", "topk") ---
    {
      snap.craned_metas = {node("cn0", 4, 8), node("cn1", 4, 8), node("cn2", 4, 8)};
      snap.partitions = {{"CPU", {"cn0", "cn1", "cn2"}}};
      algo.SetClusterSnapshot(snap);
      auto avail = [](int cores, uint64_t gib) {
        ResourceInNodeV3 r;
        r.cpu_set.cpu_count = cpu_t(cores);
        for (int c = 0; c < cores; ++c) r.cpu_set.core_ids.insert((uint32_t)c);
        r.memory_bytes = gib << 30;
        return r;
      };
      auto step = [](uint32_t id, uint32_t k, uint32_t ntasks, int cpus, uint32_t tmax) {
        auto s = std::make_unique<StepInScheduler>();
        s->step_id = id; s->node_num = k; s->ntasks = ntasks; s->ntasks_per_node_max = tmax;
        s->req_task_res_view.cpu_count = cpu_t(cpus); s->req_task_res_view.memory_bytes = 1ull << 30;
        return s;
      };
      ResourceV3 a0{{"cn0", avail(4, 8)}, {"cn1", avail(4, 8)}};                        // job 10: "fifo"
      ResourceV3 a1{{"cn0", avail(1, 8)}, {"cn1", avail(3, 8)}, {"cn2", avail(2, 8)}};  // job 11: "topk"
      auto sA = step(0, 1, 2, 1, 4), sB = step(1, 2, 4, 2, 2), sC = step(2, 1, 1, 1, 1), sT = step(0, 2, 5, 1, 4);
      std::vector<JobStepQueue> q(2);
      q[0].job_id = 10; q[0].step_res_avail = &a0; q[0].pending_steps = {sA.get(), sB.get(), sC.get()};
      q[1].job_id = 11; q[1].step_res_avail = &a1; q[1].pending_steps = {sT.get()};
      algo.SchedulePendingSteps(q);
      CHECK(algo.Ok());
      CHECK(sA->scheduled && !sB->scheduled && !sC->scheduled);                          // B does not fit: the queue stops
      CHECK(sA->craned_ids == std::vector<CranedId>{"cn0"});
      CHECK(sA->craned_task_map.at("cn0") == (std::set<uint32_t>{0, 1}));
      CHECK(sA->task_res_map.at(0).cpu_set.core_ids == std::set<uint32_t>{0} && sA->task_res_map.at(1).cpu_set.core_ids == std::set<uint32_t>{1});
      CHECK(a0.at("cn0").cpu_set.core_ids == (std::set<uint32_t>{2, 3}) && a0.at("cn0").memory_bytes == (6ull << 30));
      CHECK(sT->scheduled && sT->craned_ids == (std::vector<CranedId>{"cn2", "cn1"}));   // fewest tasks first
      CHECK(sT->craned_task_map.at("cn2") == (std::set<uint32_t>{0, 1}) && sT->craned_task_map.at("cn1") == (std::set<uint32_t>{2, 3, 4}));
      CHECK(sT->allocated_res.at("cn1").cpu_set.cpu_count == cpu_t(3));
      CHECK(a1.at("cn0").cpu_set.cpu_count == cpu_t(1) && a1.at("cn1").cpu_set.cpu_count == cpu_t(0));
    }

    // --- a running job shapes the snapshot (cost and availability) ------------------------------------------
    snap.craned_metas = {node("cn0", 2, 8), node("cn1", 2, 8)};
    snap.partitions = {{"CPU", {"cn0", "cn1"}}};
    algo.SetClusterSnapshot(snap);
    auto rn = std::make_unique<RnJobInScheduler>();
    rn->job_id = 77; rn->partition_id = "CPU"; rn->start_time = 900; rn->end_time = 1500;
    rn->allocated_res["cn0"].cpu_set.cpu_count = cpu_t(1);
    rn->allocated_res["cn0"].cpu_set.core_ids = {0};
    rn->allocated_res["cn0"].memory_bytes = 1ull << 30;
    running.push_back(std::move(rn));
    pd.clear();
    pd.push_back(job(1, 1, 100));  // cn0 has cost 500*0.5 = 250 -> cn1 (cost 0) wins
    pd.push_back(job(2, 2, 100));  // needs 2 cpus: cn0 has 1 free, cn1 has 1 free -> backfill on the min-cost node
    algo.NodeSelect(now, running, pd);
    CHECK(pd[0]->is_scheduled() && pd[0]->craned_ids[0] == "cn1");
    CHECK(pd[1]->reason == "Resource" || pd[1]->reason == "Priority");
    CHECK(pd[1]->start_time > now);   // (lazy write-back: a job backfilled for later carries no node list)
    // ... and the same cycle from the event-fed mirror instead of the running vector
    algo.MallocResourceFromNode("cn0", running[0]->job_id, running[0]->allocated_res);
    algo.SetRunningJobInfo(running[0]->job_id, running[0]->end_time);
    const std::string want0 = pd[0]->craned_ids[0];
    const TimeSec want1 = pd[1]->start_time;
    pd.clear();
    pd.push_back(job(1, 1, 100));
    pd.push_back(job(2, 2, 100));
    algo.NodeSelect(now, pd);
    CHECK(pd[0]->is_scheduled() && pd[0]->craned_ids[0] == want0 && pd[1]->start_time == want1);
    // ... and what goes on the wire for it: one record, for the job that starts now, equal to the per-job emission
    GpuNodeSelectionAlgo::WireBatch wb;
    CHECK(algo.EmitStartedResourcesWire(&wb) == 1 && wb.recs.size() == 1);
    if (wb.recs.size() == 1) {
      CHECK(algo.LastOrder()[wb.recs[0].job] == pd[0].get());
      std::string one, jtd;
      CHECK(algo.AppendResourceInNodeV3Wire(*pd[0], want0, &one) && one == wb.bytes.substr(wb.recs[0].off, wb.recs[0].len));
      // {cpu_ids [c], cpu_count 1.0, memory_bytes = request, memory_sw_bytes, gres {}}: 0A 01 cc 11 <1.0> 18 .. 20 .. 2A 00
      CHECK(one.size() > 12 && (unsigned char)one[0] == 0x0A && one[1] == 1 && (unsigned char)one[3] == 0x11 && one.back() == 0);
      CHECK(!algo.AppendResourceInNodeV3Wire(*pd[1], want0, &one));   // backfilled for later: nothing to dispatch
      CHECK(algo.AppendJobToDWire(*pd[0], 1000, "j", want0, &jtd) && jtd.size() > one.size());
    }
  }
  if (!no_gpu) {
    // ---- preemption through the adapter (tests/kat_preempt.py, scenarios P1 and P3: derivations there) -------------------
    GpuNodeSelectionAlgo algo(0);
    ClusterSnapshot snap;
    snap.craned_metas = {node("cn0", 2, 8)};
    snap.partitions = {{"CPU", {"cn0"}}};
    snap.preempt_enabled = true;
    snap.qos_preempt = {{"high", {"low"}}, {"low", {}}};
    algo.SetClusterSnapshot(snap);
    CHECK(algo.Ok());
    std::vector<std::unique_ptr<RnJobInScheduler>> rn;
    auto r0 = std::make_unique<RnJobInScheduler>();
    r0->job_id = 50; r0->partition_id = "CPU"; r0->qos = "low"; r0->qos_priority = 1; r0->start_time = 900; r0->end_time = 1500;
    r0->allocated_res["cn0"].cpu_set.cpu_count = cpu_t(2);
    r0->allocated_res["cn0"].cpu_set.core_ids = {0, 1};
    r0->allocated_res["cn0"].memory_bytes = 2ull << 30;
    rn.push_back(std::move(r0));
    std::vector<std::unique_ptr<PdJobInScheduler>> pd;
    pd.push_back(job(1, 2, 100));
    pd[0]->qos = "high"; pd[0]->qos_priority = 10; pd[0]->priority = 1.0;
    algo.NodeSelect(now, rn, pd);
    CHECK(algo.Ok());
    CHECK(pd[0]->is_scheduled() && pd[0]->start_time == now && pd[0]->craned_ids == std::vector<CranedId>{"cn0"});
    CHECK(pd[0]->preempted_jobs.size() == 1 && std::holds_alternative<RnJobInScheduler*>(pd[0]->preempted_jobs[0]) &&
          std::get<RnJobInScheduler*>(pd[0]->preempted_jobs[0]) == rn[0].get());
    CHECK(algo.LastPreemptCancel() == std::vector<job_id_t>{50} && algo.PreemptingSet() == std::set<job_id_t>{50});
    // the next cycle: job 50 still runs -> it ends at now + 1 (cpp:6553-6556) and is not cancelled a second time
    pd.clear();
    pd.push_back(job(2, 2, 100));
    pd[0]->qos = "low"; pd[0]->qos_priority = 1;
    algo.NodeSelect(now, rn, pd);
    CHECK(algo.Ok() && pd[0]->start_time == now + 1 && pd[0]->preempted_jobs.empty() && algo.LastPreemptCancel().empty());
    CHECK(algo.PreemptingSet() == std::set<job_id_t>{50});
    // a pending job placed earlier in the same cycle is preempted: reason "Preempted"
    rn.clear();
    pd.clear();
    pd.push_back(job(3, 2, 100)); pd[0]->qos = "low"; pd[0]->qos_priority = 1; pd[0]->priority = 5.0;
    pd.push_back(job(4, 2, 50)); pd[1]->qos = "high"; pd[1]->qos_priority = 10; pd[1]->priority = 1.0;
    algo.NodeSelect(now, rn, pd);
    CHECK(algo.Ok() && pd[0]->reason == "Preempted" && pd[1]->is_scheduled() && pd[1]->start_time == now);
    CHECK(pd[1]->preempted_jobs.size() == 1 && std::holds_alternative<PdJobInScheduler*>(pd[1]->preempted_jobs[0]) &&
          std::get<PdJobInScheduler*>(pd[1]->preempted_jobs[0]) == pd[0].get());
    CHECK(algo.PreemptingSet().empty());   // job 50 no longer runs: dropped from the set (cpp:6551-6553)
  }
  if (!no_gpu) {
    // ---- preemption + partitions that share a node (tests/kat_preempt.py, scenario P6): the release lowers the cost of
    // cn0 in SUB's selector only, so ALL still walks cn1 first ---------------------------------------------------------
    GpuNodeSelectionAlgo algo(0);
    ClusterSnapshot snap;
    snap.craned_metas = {node("cn0", 2, 8), node("cn1", 2, 8)};
    snap.partitions = {{"ALL", {"cn0", "cn1"}}, {"SUB", {"cn0"}}};
    snap.preempt_enabled = true;
    snap.qos_preempt = {{"high", {"low"}}, {"low", {}}};
    algo.SetClusterSnapshot(snap);
    CHECK(algo.Ok());
    std::vector<std::unique_ptr<RnJobInScheduler>> rn;
    for (int i = 0; i < 2; ++i) {
      auto r = std::make_unique<RnJobInScheduler>();
      const std::string cn = i ? "cn1" : "cn0";
      r->job_id = 50 + i; r->partition_id = "ALL"; r->qos = "low"; r->qos_priority = 1; r->start_time = 900 + 50 * i; r->end_time = i ? 1300 : 1500;
      r->allocated_res[cn].cpu_set.cpu_count = cpu_t(2);
      r->allocated_res[cn].cpu_set.core_ids = {0, 1};
      r->allocated_res[cn].memory_bytes = 2ull << 30;
      rn.push_back(std::move(r));
    }
    std::vector<std::unique_ptr<PdJobInScheduler>> pd;
    pd.push_back(job(1, 2, 100)); pd[0]->partition_id = "SUB"; pd[0]->qos = "high"; pd[0]->qos_priority = 10; pd[0]->priority = 1.0;
    pd.push_back(job(2, 2, 100)); pd[1]->partition_id = "ALL"; pd[1]->qos = "low"; pd[1]->qos_priority = 1; pd[1]->priority = 1.0;
    algo.NodeSelect(now, rn, pd);
    CHECK(algo.Ok());
    CHECK(pd[0]->is_scheduled() && pd[0]->start_time == now && pd[0]->craned_ids == std::vector<CranedId>{"cn0"});
    CHECK(pd[0]->preempted_jobs.size() == 1 && std::holds_alternative<RnJobInScheduler*>(pd[0]->preempted_jobs[0]) &&
          std::get<RnJobInScheduler*>(pd[0]->preempted_jobs[0]) == rn[0].get());
    CHECK(pd[1]->start_time == 1300 && pd[1]->reason == "Resource");   // (on cn1; a later start carries reason / start / end only)
    CHECK(algo.LastPreemptCancel() == std::vector<job_id_t>{50} && algo.PreemptingSet() == std::set<job_id_t>{50});
  }
  printf("%s\n", g_fail ? "FAIL" : "ok");
  return g_fail != 0;
}

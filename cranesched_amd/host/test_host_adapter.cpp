// Drives GpuNodeSelectionAlgo the way JobScheduler::ScheduleThread_ drives SchedulerAlgo
// (src/CraneCtld/JobScheduler.cpp:1375-1447): build PdJobInScheduler objects, call NodeSelect once,
// read start_time / craned_ids / allocated_res / reason back.  Expected values are the hand-derived
// known answers of tests/kat.py (scenarios A, B and F).
//   test_host_adapter            -> needs an MI355X, exit 0 on success
//   test_host_adapter --no-gpu   -> checks the loud "no device" behaviour instead
#include <cstdio>
#include <cstring>
#include <string>

#include "NodeSelectionAlgo.h"

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

int main(int argc, char** argv) {
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
    CHECK(!pd[1]->craned_ids.empty() && pd[1]->start_time > now);
  }
  printf("%s\n", g_fail ? "FAIL" : "ok");
  return g_fail != 0;
}

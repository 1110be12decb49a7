// Exact emulation of std::priority_queue<node_info> as built by libstdc++ (GCC 11 bits/stl_heap.h:
// __push_heap / __adjust_heap / __pop_heap).  The reference keeps its top-k node candidates in two
// such queues (src/CraneCtld/JobScheduler.cpp:6157-6169) ordered by
//     bool node_info::operator<(other) const { return ntasks_on_node > other.ntasks_on_node; }
// so WHICH of several equal-capacity entries is evicted (:6238-6241, :6290-6293) and the order in
// which tasks are handed out (:6305-6325) are artefacts of the heap layout.  The engine
// reproduces that layout move for move; tests/test_pq_emul.py checks this file against the real
// std::priority_queue on the CPU.
#pragma once
#include "res_dev.h"

namespace cns {

struct HeapEnt {       // node_info {ntasks_on_node, res, node_state}
  int ntasks;          // ntasks_on_node
  u32 p;               // partition-local slot of the node
  u32 node;            // dense node index
  u32 pad;
  double cost;         // the node's cost when it was selected (for the later cost update)
  Res res;             // total / window-min resource the allocation is cut from
};

// comp(a, b) of std::less<node_info>: a < b  <=>  a.ntasks_on_node > b.ntasks_on_node
CNS_HD bool heap_comp(const HeapEnt& a, const HeapEnt& b) { return a.ntasks > b.ntasks; }

// std::__push_heap(first, holeIndex, topIndex, value, comp)
CNS_HD void heap_push_up(HeapEnt* first, int holeIndex, int topIndex, const HeapEnt& value) {
  int parent = (holeIndex - 1) / 2;
  while (holeIndex > topIndex && heap_comp(first[parent], value)) {
    first[holeIndex] = first[parent];
    holeIndex = parent;
    parent = (holeIndex - 1) / 2;
  }
  first[holeIndex] = value;
}

// priority_queue::push: c.push_back(x); std::push_heap(c.begin(), c.end()).
// `len` is the size AFTER the push; first[len-1] already holds x.
CNS_HD void pq_push(HeapEnt* first, int len) {
  HeapEnt value = first[len - 1];
  heap_push_up(first, len - 1, 0, value);
}

// std::__adjust_heap(first, holeIndex, len, value, comp)
CNS_HD void heap_adjust(HeapEnt* first, int holeIndex, int len, const HeapEnt& value) {
  const int topIndex = holeIndex;
  int secondChild = holeIndex;
  while (secondChild < (len - 1) / 2) {
    secondChild = 2 * (secondChild + 1);
    if (heap_comp(first[secondChild], first[secondChild - 1])) secondChild--;
    first[holeIndex] = first[secondChild];
    holeIndex = secondChild;
  }
  if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
    secondChild = 2 * (secondChild + 1);
    first[holeIndex] = first[secondChild - 1];
    holeIndex = secondChild - 1;
  }
  heap_push_up(first, holeIndex, topIndex, value);
}

// priority_queue::pop: std::pop_heap(c.begin(), c.end()); c.pop_back().
// `len` is the size BEFORE the pop; afterwards the queue is first[0..len-1) and the removed top
// sits in first[len-1].
CNS_HD void pq_pop(HeapEnt* first, int len) {
  if (len > 1) {
    HeapEnt value = first[len - 1];
    first[len - 1] = first[0];
    heap_adjust(first, 0, len - 1, value);
  }
}

}  // namespace cns

// Internal layout of a graph plan, shared by graph.hip (host-built plans, SpMM) and graphdev.hip (plans built
// on the device from (row, col) pairs).
#pragma once
#include "common.hpp"

namespace mmssl {

struct DirPlan {
  int32_t rows = 0, cols = 0;
  int64_t nnz = 0;
  int32_t* rowptr = nullptr;  // [rows+1]   (device)
  Edge* edges = nullptr;      // [nnz]
  int4* gitems = nullptr;     // group items  {row, beg, end, -1}
  int4* witems = nullptr;     // wave items   {row, beg, end, slot|-1}
  int4* multi = nullptr;      // multi rows   {row, first_slot, n_slots, 0}
  int32_t* slot2multi = nullptr;  // [n_slots] -> index into `multi`
  int64_t n_g = 0, n_w = 0, n_multi = 0, n_slots = 0;
  // device-built plans: the real item counts {n_g, n_w, n_multi, n_slots, nnz} live HERE (device int32[8]); the host
  // fields above then are CAPACITIES (launch grids and workspaces are sized by them)
  int32_t* dyn = nullptr;
};


void free_dir(DirPlan& p);

}  // namespace mmssl

struct mmssl_graph {
  mmssl::DirPlan fwd, bwd;
};

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
  // XCD-banded group items (host-built plans whose rows mostly reference ONE of kBands column bands): gitems is laid out
  // band-major, bands[x] .. bands[x + 1] are the items of band x (device int32[kBands + 1]); block b takes its items from
  // band b % 8 - the XCD it is observed to run on - so that an XCD's L2 holds one band of the gathered table instead of
  // all of it. band_max = the longest band (sizes the grid). nullptr: the flat degree-sorted list.
  int32_t* bands = nullptr;
  int64_t band_max = 0;
  // ... and wave blocks (4 wave items each: the slices of one heavy row, or four light rows): hardware block b processes
  // wave block wmap[b], chosen at plan time from the blocks of band b % 8 (heaviest first; an exhausted band takes over
  // blocks of the fullest one). nullptr: block b processes wave block b.
  int32_t* wmap = nullptr;
  double band_score = 0.0;      // fraction of the edges that fall into their row's dominant band
};

constexpr int kBands = 8;


void free_dir(DirPlan& p);

}  // namespace mmssl

struct mmssl_graph {
  mmssl::DirPlan fwd, bwd;
};

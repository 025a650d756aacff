// kernels.hpp — hand-written HIP kernels (gfx950 / CDNA4) of the parallel-MMA hot path.
//
// One kernel launch sweeps ALL BDDs for a whole pass (the reference launches 3 kernels per hop,
// bdd_cuda_parallel_mma.cu:207-257,301-346).  A *pack* of BDDs is walked hop by hop by one
// wavefront (narrow packs, 64 threads, barrier-free) or one workgroup (wide packs):
//   - node words / potentials / layer costs are hop-major SoA inside the pack, so every global
//     access of a wave is a contiguous stream (coalesced);
//   - the frontier (cost-from-root of the current and next hop, cost-from-terminal of the next
//     hop) lives in LDS; children are addressed by their local index inside the next hop;
//   - the per-layer min-marginal is a segmented minimum over the lanes of the layer (position inside the layer and layer index
//     come from the node word): a DPP pair for layers of <= 2 nodes, one LDS slot per layer (ds_min) for wider ones;
//   - no MFMA: this is an HBM-bound gather/scan (2 flops per 4-8 bytes).
//
// Arithmetic order follows the reference exactly (SURVEY.md §8 a'):
//   m0 = (F[u] + lo) + T[lo(u)],  m1 = (F[u] + hi) + T[hi(u)]          bdd_cuda_parallel_mma.cu:83-84
//   mm = finite(m0) && finite(m1) ? omega * (m1 - m0) : 0              :36-39
//   lo' = (lo + min(mm,0)) + delta[2v],  hi' = (hi + min(-mm,0)) + delta[2v+1]   :191-197, :286-287
//   F[child] = min(F[child], F[u] + cost')                             :194-198
//   T[u] = min(hi' + T[hi(u)], lo' + T[lo(u)])                         :292
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "layout.hpp"

// The kernels by family (round 5: one 4 000-line header before); order matters — each part uses what the ones before it define.
#include "kernels/common.hpp"
#include "kernels/narrow.hpp"
#include "kernels/resident.hpp"
#include "kernels/narrow2.hpp"
#include "kernels/narrow3.hpp"
#include "kernels/wide.hpp"
#include "kernels/exchange.hpp"
#include "kernels/small.hpp"
#include "kernels/elementwise.hpp"

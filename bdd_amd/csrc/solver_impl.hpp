// solver_impl.hpp — the solver object behind the C-ABI (include/bdd_mma.h): template SolverT<REAL>, instantiated once per precision in its
// own translation unit (solver_f32.hip / solver_f64.hip, so that the two compile side by side; solver_base.hip has the precision-free part).
// gfx950 only; there is no CPU fallback: every compute entry point fails with BDDMMA_ERR_DEVICE when no HIP device is usable.
//
// Mirrors LPMP::bdd_cuda_parallel_mma<REAL> / bdd_cuda_base<REAL>
// (reference: src/bdd_solver/bdd_cuda_parallel_mma.cu, src/bdd_solver/bdd_cuda_base.cu).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include "../../include/bdd_mma.h"
#include "kernels.hpp"
#include "layout.hpp"
#include "solver.hpp"

namespace bddmma {

// one step of a host spin-wait (the polled bounds, run_solver's reader): the core's pause / yield hint
static inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __asm__ __volatile__("pause");
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#else
    std::this_thread::yield();
#endif
}


#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                                   \
            return BDDMMA_ERR_DEVICE;                                                                  \
        }                                                                                              \
    } while (0)

static inline uint32_t cdiv(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------
// accumulator type of the exchange kernel's LDS tile (kernels.hpp: k_exchange_reduce); experiments: -DBDDMMA_EX_ACC=REAL
#ifndef BDDMMA_EX_ACC
#define BDDMMA_EX_ACC double
#endif
template <typename REAL>
struct SolverT final : SolverBase {
    // device buffers
    uint32_t* d_nwords = nullptr;       // distinct pack word sequences (layout.hpp: narrow_words_unique)
    uint32_t* d_pack_word_off = nullptr;
    uint32_t n_nwords = 0;
    uint64_t* d_wwords = nullptr;
    REAL *d_F = nullptr, *d_T = nullptr, *d_lohi = nullptr;  // d_lohi: {lo, hi} per layer, interleaved
    REAL* d_lo = nullptr;  // = d_lohi     (stride 2)
    REAL* d_hi = nullptr;  // = d_lohi + 1 (stride 2)
    int32_t *d_var = nullptr, *d_bdd = nullptr, *d_nbdds = nullptr;
    uint32_t *d_var_ptr = nullptr, *d_var_layers = nullptr, *d_root_slot = nullptr;
    REAL* d_delta_var = nullptr;    // 2V: the solver's deferred delta_lo_hi_ (normalised), per variable; rebuilt lazily, see delta_var_valid
    bool delta_var_valid = true;    // false after an exchange that only wrote the broadcast pairs (d_delta_lay)
    REAL* d_delta_lay = nullptr;    // 2L: the same, broadcast to binned entry order (what the sweeps read)
    REAL* d_mm_binned = nullptr;    // L : deferred min-marginal differences in binned entry order
    REAL *d_delta_c = nullptr, *d_delta_lay_c = nullptr;  // scratch for the explicit forward_mm/backward_mm API
    uint16_t* d_bvar = nullptr;
    uint32_t *d_evar = nullptr, *d_lpos = nullptr, *d_vpos = nullptr, *d_bin_ptr = nullptr;
    uint32_t *d_pack_group_ptr = nullptr, *d_grp_layer_off = nullptr, *d_grp_hop_end = nullptr;
    uint32_t *d_quad_round_ptr = nullptr, *d_cs_ptr = nullptr, *d_cs_entry = nullptr;
    uint16_t* d_cs_slot = nullptr;
    uint32_t wpb = 1;
    bool entry_by_var = false;  // entries ordered by (variable, bdd): exchange = k_exchange_byvar
    bool exch_small = false, exch_medium = false;
    // exchange as a fixed schedule (kernels.hpp: k_exchange_seg; layout.hpp: SegExchange) where every bin fits it
    bool use_seg = false;
    uint32_t *d_seg_bin = nullptr, *d_seg_thr = nullptr;
    uint16_t* d_seg_perm = nullptr;
    uint32_t seg_lds = 0, seg_tile_off = 0, seg_cnt_off = 0, seg_groups = 0;
    bool big = false;  // some array reaches 4 GiB (or variant_flags bit 14, for the tests): kernels.hpp DevPtrs::big
    uint32_t opts_variant = 0;  // bddmma_options.variant_flags (A/B switches of kernel variants)
    uint32_t n_cus = 256, lds_cu = 160 * 1024;  // hipDeviceProp of `device` (init)
    bool narrow_seg = false;  // some narrow pack has layers wider than two nodes: seg_min2 goes through LDS and needs scratch
    uint32_t vars_per_bin = 0, n_bins = 0, stage_cap = 0, stage_lds = 0, exch_lds = 0, n_narrow_layers = 0;
    double *d_lb_partial = nullptr, *d_lb = nullptr;
    double* h_lb = nullptr;  // pinned, device-visible: the reduce kernel writes the bound straight into host memory
    // h_lb[0]: lower_bound(); h_lb[1], h_lb[2]: the slots of lower_bound_enqueue; each with a sequence word the reduce kernel writes behind
    // the value, polled by the host (wait_bound)
    uint64_t *h_lb_seq = nullptr, *d_lb_seq = nullptr;
    uint64_t lb_seq_next = 0, lb_seq_expect[3] = {0, 0, 0};
    // the bound is a function of the costs-to-terminal, which only launch_bwd() writes: a second lower_bound() without a backward
    // sweep in between (L-BFGS reads the bound at the end of an iteration and again before its step search) costs nothing
    bool lb_cached = false;
    double lb_cache = 0.0;
    bool lb_slot_known[2] = {false, false}, lb_slot_enqueued[2] = {false, false};  // lower_bound_enqueue / _fetch: h_lb[1], h_lb[2]
    double lb_slot_value[2] = {0.0, 0.0};
    uint64_t lb_gen = 0, lb_slot_gen[2] = {0, 0};       // lb_gen counts the backward launches: a fetched slot of the current generation is the cached bound
    // device-resident run_solver (kernels.hpp: run_ctl_step)
    RunCtl* d_run_ctl = nullptr;
    RunHost *h_run = nullptr, *d_run_host = nullptr;  // pinned + its device address
    const uint32_t* run_stop = nullptr;               // &d_run_ctl->stop while run_plain() queues iterations, nullptr otherwise
    uint32_t run_iter = 0;                            // index of the iteration being queued by run_plain() (kernels.hpp: DevPtrs::run_iter)
    RunGate gate() const { return RunGate{run_stop, run_iter}; }
    RunStep run_step{};                               // {partials, count, ctl, host}: what the launch that ends an iteration gets
    REAL* d_x_layer = nullptr;    // net_solver_costs x = (hi - lo) + deferred mm in layer order, written by the backward solve sweeps once an L-BFGS wrapper uses them
    bool x_layer_valid = false;   // true after a backward solve sweep, false after anything else that writes arc costs or deferred values
    // every call that changes arc costs other than through a solve sweep: both sweep states and the x view are stale, and an L-BFGS wrapper's
    // pending bound is no longer "the bound of the current costs" (SolverBase::cost_epoch)
    void costs_changed()
    {
        fwd_valid = bwd_valid = x_layer_valid = false;
        ++cost_epoch;
    }
    CostQuot* d_cost_q = nullptr;      // update_costs: per-variable quotients (kernels.hpp: k_cost_quotients)
    uint8_t* d_cost_flags = nullptr;
    uint32_t* d_counts = nullptr;
    REAL *d_tmp0 = nullptr, *d_tmp1 = nullptr;        // per-layer scratch (min-marginals, sorted outputs)
    char* d_sol = nullptr;
    struct PackBufs {
        uint32_t *pack_hop_ptr = nullptr, *hop_node_off = nullptr, *hop_layer_off = nullptr;
        uint8_t* pack_steps = nullptr;
        uint16_t* hop_root = nullptr;  // narrow and wide packs (layout.hpp: PackSet::hop_root)
        uint32_t n_packs = 0;
    } nb_, wb_, hb_;  // narrow, wide, huge packs
    uint32_t wide_lds = 0, wide_threads = 256, wide_npt = 1;
    bool n3_nt = false;  // third-generation sweeps: the instantiation with non-temporal loads (double beyond the Infinity Cache's reach)
    bool beyond_cache = false;  // the arrays exceed what the Infinity Cache helps with (640 MiB)
    bool n12_nt = false;  // ... and of the first- (float) / second-generation (double) sweeps that run beyond 16 M slots: packs of 128 slots, 4 / 8 per workgroup
    uint32_t nt_potentials = 0;  // PackDev::nt_potentials (kernels.hpp: hop_store): double, footprint several times the Infinity Cache
    bool mixed = false;      // narrow (streaming) and wide solve sweeps in one launch (kernels.hpp: k_fwd_mixed / k_bwd_mixed)
    // Measured on the knapsack benchmark (3 604 narrow + 389 wide packs): backward 44.7 -> 37.4 us in one launch.  The forward sweeps
    // did not gain at first (56.1 -> 58.6 us: both kinds were bound by same-address LDS pushes into the sink entries); with those
    // pushes gone the one launch wins there too: 32.7 -> 26.3 us (float), 37.3 -> 30.2 us (double), 14.5 k -> 16.1 k it/s.
    bool mixed_fwd = true;   // variant_flags bit 1: forward narrow / wide solve sweeps as two launches
    uint32_t mixed_npt = 1, mixed_lds = 0;
    // (Running the wide launch on a second stream next to the narrow one was measured and dropped: the event fork / join costs ~10 us
    // per pass on this platform, more than the overlap returns — knapsack benchmark 8 990 -> 8 261 it/s.)
    // resident sweeps of the narrow packs (kernels.hpp: k_fwd_res / k_bwd_res)
    uint32_t *d_pack_hdr = nullptr, *d_quad_hdr = nullptr;
    bool use_res = false;
    uint32_t res_ns = 0, res_nl = 0, res_lds = 0;
    // second generation (kernels.hpp: k_fwd_res2 / k_bwd_res2): per-lane records of ready-made LDS offsets, derived from the layout here
    bool use_res2 = false;
    uint32_t *d_res2_rec = nullptr, *d_res2_rec_off = nullptr;
    uint32_t res2_n_words = 0, res2_lds = 0, res2_ns = 0, res2_nl = 0;
    // streaming solve sweeps on per-lane records (kernels.hpp: k_fwd_narrow2 / k_bwd_narrow2)
    bool use_narrow2 = false, narrow_gen = false;  // narrow_gen: layers wider than two nodes or staggered packs -> the kernels' general form
    uint32_t *d_srec = nullptr, *d_srec_off = nullptr;
    // the streaming sweeps of the narrow packs start from the resident headers where those hold (one stage group per pack, one round per quad, no
    // staggered packs; layout.hpp: struct Resident) — variant_flags bit 16: from the hop / group / round tables as before
    bool res_hdr_ok = false;
    const uint32_t* n2_hdr_pack() const { return res_hdr_ok ? d_pack_hdr : nullptr; }
    const uint32_t* n2_hdr_quad() const { return res_hdr_ok ? d_quad_hdr : nullptr; }
    uint32_t srec_words = 0;
    // instances that fit one workgroup: whole iterations in one launch (kernels/small.hpp: k_iterate_small)
    bool small_ok = false, small_rl = false;
    int small_nw = 1;
    uint32_t small_lds = 0;
    SmallDev small{};
    uint32_t* d_small_lds = nullptr;
    // streaming solve sweeps, third generation: a lane per layer (kernels/narrow3.hpp: k_fwd_narrow3 / k_bwd_narrow3)
    bool use_narrow3 = false;
    uint32_t *d_lrec = nullptr, *d_lrec_off = nullptr;
    uint32_t lrec_words = 0;
    uint32_t huge_pack_width = 0;
    unsigned char* d_huge_scratch = nullptr;  // frontier arrays of the huge packs (global memory instead of LDS)

    std::vector<void*> allocs;

    ~SolverT() override
    {
        if (device >= 0) (void)hipSetDevice(device);
        for (void* p : allocs) (void)hipFree(p);
        if (h_lb) (void)hipHostFree(h_lb);
        if (h_lb_seq) (void)hipHostFree(h_lb_seq);
        if (h_run) (void)hipHostFree(h_run);
        if (d_run_ctl) (void)hipFree(d_run_ctl);
        for (auto& e : ev_pool) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
        if (ev_t0) (void)hipEventDestroy(ev_t0);
        if (ev_t1) (void)hipEventDestroy(ev_t1);
        if (stream) (void)hipStreamDestroy(stream);
    }

    // Arena: the solver's arrays are carved out of few large allocations instead of one hipMalloc each (see arena_chunk below).
    struct Arena { char* base = nullptr; uint64_t size = 0, used = 0; };
    std::vector<Arena> arenas;
    uint64_t arena_chunk = 0;   // bytes per arena allocation (0: one hipMalloc per array, as before round 5)
    uint64_t arena_skew = 0;    // extra bytes between consecutive arrays
    template <typename T>
    int dalloc(T** p, uint64_t n)
    {
        // + 1 KiB: the resident sweeps copy in whole 1 KiB pieces and may read up to 1008 bytes past a pack's range (kernels.hpp: wave_copy_to_lds)
        const uint64_t bytes = std::max<uint64_t>(n, 1) * sizeof(T) + 1024;
        if (arena_chunk == 0) {
            HIPCHK(hipMalloc((void**)p, bytes));
            allocs.push_back(*p);
            dev_bytes += bytes;
            dev_alloc_bytes += bytes;
            return BDDMMA_OK;
        }
        const uint64_t need = (bytes + 4095) / 4096 * 4096 + arena_skew;
        if (arenas.empty() || arenas.back().used + need > arenas.back().size) {
            Arena a;
            a.size = std::max(arenas.empty() ? arena_chunk : std::min<uint64_t>(arena_chunk, 64ull << 20), need);   // what comes after the first chunk is small
            HIPCHK(hipMalloc((void**)&a.base, a.size));
            allocs.push_back(a.base);
            arenas.push_back(a);
            dev_alloc_bytes += a.size;
        }
        Arena& a = arenas.back();
        *p = reinterpret_cast<T*>(a.base + a.used);
        a.used += need;
        dev_bytes += bytes;
        return BDDMMA_OK;
    }
    struct DevField { const void* ptr = nullptr; uint64_t count = 0; };
    DevField dev_fields[LAYOUT_ARRAY_IDS];  // layout array id (layout.hpp: visit_layout_arrays) -> where it lives on the device
    template <typename T>
    int upload(T** p, const std::vector<T>& h, int id = 0)
    {
        int rc = dalloc(p, h.size());
        if (rc) return rc;
        if (!h.empty()) HIPCHK(hipMemcpyAsync(*p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, stream));
        if (id) dev_fields[id] = DevField{*p, h.size()};
        return BDDMMA_OK;
    }
    int download_layout(HostLayout& H) override
    {
        HIPCHK(hipSetDevice(device));
        H = HostLayout();
        set_layout_scalars(H, lay_scalars);
        hipError_t e = hipSuccess;
        visit_layout_arrays(H, [&](int id, auto& vec) {
            using T = typename std::remove_reference_t<decltype(vec)>::value_type;
            if (id == 36) { vec.assign(nodes_per_hop.begin(), nodes_per_hop.end()); return; }
            if (id == 37) { vec.assign(layers_per_hop.begin(), layers_per_hop.end()); return; }
            const DevField& f = dev_fields[id];
            vec.resize(f.count);
            if (f.count && e == hipSuccess) e = hipMemcpyAsync(vec.data(), f.ptr, f.count * sizeof(T), hipMemcpyDeviceToHost, stream);
        });
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        HIPCHK(e);
        return BDDMMA_OK;
    }
    int upload_packs(PackBufs& b, const PackSet& ps, int id0)
    {
        b.n_packs = ps.n_packs();
        int rc;
        if ((rc = upload(&b.pack_hop_ptr, ps.pack_hop_ptr, id0))) return rc;
        if ((rc = upload(&b.hop_node_off, ps.hop_node_off, id0 + 1))) return rc;
        if ((rc = upload(&b.hop_layer_off, ps.hop_layer_off, id0 + 2))) return rc;
        if ((rc = upload(&b.pack_steps, ps.pack_steps, id0 + 3))) return rc;
        return BDDMMA_OK;
    }

    int init_from_layout(const HostLayout& L, const bddmma_options* opts) override { return init(L, opts); }
    int init(const HostLayout& L, const bddmma_options* opts)
    {
        HIPCHK(hipSetDevice(device));
        {
            ChipInfo chip;
            std::string e;
            if (query_chip(device, &chip, e)) { err = e; return BDDMMA_ERR_DEVICE; }
            n_cus = chip.n_cus;
            lds_cu = chip.lds_bytes;
        }
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreate(&ev_t0));
        HIPCHK(hipEventCreate(&ev_t1));
        n_vars = L.n_vars; n_bdds = L.n_bdds; n_layers = L.n_layers; n_hops = L.n_hops;
        n_input_nodes = L.n_input_nodes; n_slots = L.n_slots; pack_width = L.pack_width;
        wide_pack_width = L.wide_pack_width;
        nodes_per_hop = L.nodes_per_hop; layers_per_hop = L.layers_per_hop;
        h_nbdds = L.num_bdds_per_var; h_var_ptr = L.var_ptr;
        h_layer_var = L.layer_var; h_layer_bdd = L.layer_bdd;
        lay_scalars = layout_scalars(L);
        if (opts) saved_opts = *opts;
        deterministic = opts && opts->deterministic;
        {
            // Placement (round 5, tools/placement_probe.py, profiles/r05_placement.txt).  With one hipMalloc per array the iteration rate of
            // the SAME instance depends on where the arrays happened to land: the 10.5 M-node float instance runs at 8 810, 8 200, 7 500 or
            // 6 960 it/s from one solver object to the next in one process (stable for the life of an object; the sweeps alone always take
            // 38.7 / 39.8 us, inside the iteration loop 41 ... 57 us: what varies is how much of the 465 MB working set the 256 MiB Infinity
            // Cache keeps between launches) — the "noisy boxes" of rounds 3-4.  Carved out of ONE allocation, 4 KiB-aligned back to back,
            // the arrays gave 8 800-8 840 it/s in 40 of 40 solver objects on three boxes (a deliberate 1 MiB + 64 KiB between arrays: 8 000
            // in 10 of 10).  Beyond the cache's reach the contiguous placement is the consistently SLOWER one (21 M nodes 3 580 against
            // 3 590-3 800, 42 M 1 650 against 1 640-1 840 it/s), so the arena is used while the solver's arrays are expected to stay below
            // ~640 MiB; 1 M / 4.2 M nodes and double at 4.2 M: no difference.
            const uint64_t est = (uint64_t)n_slots * 2 * sizeof(REAL) + (uint64_t)n_layers * (5 * sizeof(REAL) + 24) + (uint64_t)n_vars * 24 +
                                 L.narrow_words_unique.size() * 4 + L.wide_words.size() * 8;
            // ~1.6 x: measured 1.45 x on the headline instance.  Small instances get a small chunk (rounded up to 2 MiB; it was floored at
            // 32 MiB through round 5: a farm of per-subproblem solvers then held 32 MiB each while bddmma_device_bytes reported ~1 MB).
            const uint64_t want = est + est / 2 + est / 8 + std::min<uint64_t>(16ull << 20, est / 2 + (1ull << 20));
            if (want <= (640ull << 20)) arena_chunk = (want + (2ull << 20) - 1) / (2ull << 20) * (2ull << 20);
#ifdef BDDMMA_EXPERIMENTAL  // make EXPERIMENTAL=1: BDDMMA_EXP_ARENA="<chunk MiB>,<bytes between arrays>" (0: one hipMalloc per array) — tools/exp_r05_aa.sh ... _ad.sh
            if (const char* ar = std::getenv("BDDMMA_EXP_ARENA")) {
                unsigned long long mib = 0, skew = 0;
                if (std::sscanf(ar, "%llu,%llu", &mib, &skew) >= 1) { arena_chunk = mib << 20; arena_skew = skew; }
            }
#endif
        }
        // Arrays of 4 GiB and more (>= 512 M slots or 256 M layers in double; the reference indexes nodes with int: 2^31).  The narrow sweeps
        // and the exchange address F / T / {lo, hi} and the entry arrays relative to their pack / bin (HopWindow, exchange_reduce_body) and
        // stage with 64-bit addresses then (DevPtrs::big); what still carries absolute 32-bit byte offsets is refused for such instances
        // below: wide and huge packs, the deterministic and by-variable exchanges, resident sweeps (small instances anyway), the L-BFGS
        // wrapper's vector passes.
        big = (uint64_t)n_slots * sizeof(REAL) >= 0xFFFF0000ull || 2ull * n_layers * sizeof(REAL) >= 0xFFFF0000ull || (opts && (opts->variant_flags & 0x4000u));
        if ((uint64_t)n_slots * sizeof(REAL) >= 0xFFFF0000ull || 2ull * n_layers * sizeof(REAL) >= 0xFFFF0000ull) {
            if (L.wide.n_packs() || L.huge.n_packs() || deterministic || L.ex.entry_by_var) {
                err = "instance too large for the 32-bit buffer offsets of this configuration: " + std::to_string(n_slots) + " node slots, " +
                      std::to_string(n_layers) + " layers of " + std::to_string(sizeof(REAL)) +
                      "-byte values (arrays of 4 GiB and more are supported for narrow packs with the binned exchange)";
                return BDDMMA_ERR_UNSUPPORTED;
            }
        }
        int rc;
        if ((rc = upload(&d_nwords, L.narrow_words_unique, 1))) return rc;
        n_nwords = (uint32_t)L.narrow_words_unique.size();
        if ((rc = upload(&d_pack_word_off, L.narrow_word_off, 2))) return rc;
        if ((rc = upload(&d_wwords, L.wide_words, 3))) return rc;
        if ((rc = upload(&d_var, L.layer_var, 4))) return rc;
        if ((rc = upload(&d_bdd, L.layer_bdd, 5))) return rc;
        if ((rc = upload(&d_nbdds, L.num_bdds_per_var, 6))) return rc;
        if ((rc = upload(&d_var_ptr, L.var_ptr, 7))) return rc;
        if ((rc = upload(&d_var_layers, L.var_layers, 8))) return rc;
        if ((rc = upload(&d_root_slot, L.bdd_root_slot, 9))) return rc;
        if ((rc = upload_packs(nb_, L.narrow, 10))) return rc;
        if (L.narrow.hop_root.size() + 1 != L.narrow.hop_node_off.size() && !L.narrow.hop_node_off.empty()) {
            err = "narrow hop_root table does not match the hop records";
            return BDDMMA_ERR_INVALID_ARGUMENT;
        }
        if ((rc = upload(&nb_.hop_root, L.narrow.hop_root, 38))) return rc;
        if ((rc = upload_packs(wb_, L.wide, 14))) return rc;
        if (L.wide.hop_root.size() + 1 != L.wide.hop_node_off.size() && !L.wide.hop_node_off.empty()) {
            err = "wide hop_root table does not match the hop records";
            return BDDMMA_ERR_INVALID_ARGUMENT;
        }
        if ((rc = upload(&wb_.hop_root, L.wide.hop_root, 39))) return rc;
        if ((rc = upload_packs(hb_, L.huge, 18))) return rc;
        huge_pack_width = L.huge_pack_width;
        if (hb_.n_packs && (rc = dalloc(&d_huge_scratch, (size_t)hb_.n_packs * wide_lds_bytes(sizeof(REAL), huge_pack_width, true)))) return rc;
        wide_slot_base = L.narrow_slots;
        if ((rc = dalloc(&d_F, n_slots))) return rc;
        if ((rc = dalloc(&d_T, n_slots))) return rc;
        if ((rc = dalloc(&d_lohi, 2 * n_layers + 2))) return rc;
        d_lo = d_lohi;
        d_hi = d_lohi + 1;
        if ((rc = dalloc(&d_mm_binned, n_layers))) return rc;
        if ((rc = dalloc(&d_delta_lay, 2 * n_layers))) return rc;
        if ((rc = dalloc(&d_delta_lay_c, 2 * n_layers))) return rc;
        if ((rc = upload(&d_evar, L.ex.evar, 22))) return rc;
        if ((rc = upload(&d_bvar, L.ex.bvar, 23))) return rc;
        if ((rc = upload(&d_lpos, L.ex.lpos, 24))) return rc;
        if ((rc = upload(&d_vpos, L.ex.vpos, 25))) return rc;
        if ((rc = upload(&d_bin_ptr, L.ex.bin_ptr, 26))) return rc;
        if ((rc = upload(&d_pack_group_ptr, L.ex.pack_group_ptr, 27))) return rc;
        if ((rc = upload(&d_grp_layer_off, L.ex.grp_layer_off, 28))) return rc;
        if ((rc = upload(&d_grp_hop_end, L.ex.grp_hop_end, 29))) return rc;
        if ((rc = upload(&d_quad_round_ptr, L.ex.quad_round_ptr, 30))) return rc;
        if ((rc = upload(&d_cs_ptr, L.ex.cs_ptr, 31))) return rc;
        if ((rc = upload(&d_cs_entry, L.ex.cs_entry, 32))) return rc;
        if ((rc = upload(&d_cs_slot, L.ex.cs_slot, 33))) return rc;
        wpb = L.ex.waves_per_block;
        entry_by_var = L.ex.entry_by_var;
        {   // balance of the narrow packs' hop counts over contiguous eighths of the pack sequence (see xcd_chunk)
            const uint32_t P = L.narrow.n_packs();
            if (P >= 64) {
                uint64_t part[8] = {0, 0, 0, 0, 0, 0, 0, 0}, total = 0;
                const uint32_t per = (P + 7) / 8;
                for (uint32_t p = 0; p < P; ++p) {
                    const uint32_t h = L.narrow.pack_hop_ptr[p + 1] - L.narrow.pack_hop_ptr[p];
                    part[std::min<uint32_t>(7, p / per)] += h;
                    total += h;
                }
                uint64_t mx = 0;
                for (uint64_t v : part) mx = std::max(mx, v);
                if (mx * 8 * 10 > total * 11) xcd_chunk_auto = 32;
            }
        }
        for (uint8_t st : L.narrow.pack_steps) narrow_seg = narrow_seg || st >= 2;
        vars_per_bin = L.ex.vars_per_bin; n_bins = L.ex.n_bins; stage_cap = L.ex.stage_cap;
        n_narrow_layers = L.ex.grp_layer_off.empty() ? 0 : L.ex.grp_layer_off.back();
        stage_lds = L.ex.waves_per_block * stage_cap * 2 * (uint32_t)sizeof(REAL);
        exch_lds = vars_per_bin * 2 * (uint32_t)sizeof(double);  // accumulators are double for both precisions
        // LDS of a narrow solve workgroup: the dynamic staging area + the kernel's static arrays (frontier F x2 and T per wave, the hop
        // offset windows) — k_fwd_narrow is the larger of the two.  Above the 64 KiB a launch gets by default the kernels need the
        // attribute (pack_width 256, double, 4 waves per block is ~67 KiB); above what a CU has the options are refused here,
        // not at the first sweep.
        const uint32_t narrow_static = L.ex.waves_per_block * (3 * (pack_width + 2) * (uint32_t)sizeof(REAL) + 2 * 64 * 4 + 2) + seg_bytes(L.ex.waves_per_block);
        if (exch_lds > lds_cu - 1024 || stage_lds + narrow_static > lds_cu - 1024) {
            err = "vars_per_bin / stage_cap / waves_per_block need more LDS than a CU has (" + std::to_string(stage_lds + narrow_static) + " B per sweep workgroup)";
            return BDDMMA_ERR_INVALID_ARGUMENT;
        }
        if (stage_lds + narrow_static > 64 * 1024) {
#define SET_N1(K_) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&K_), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(stage_lds + seg_bytes(L.ex.waves_per_block))));
#define SET_N(R_, W_) SET_N1((k_fwd_narrow<REAL, R_, FWD_SOLVE, W_>)) SET_N1((k_bwd_narrow<REAL, R_, BWD_SOLVE, W_>)) SET_N1((k_fwd_narrow<REAL, R_, FWD_SOLVE, W_, false>)) SET_N1((k_bwd_narrow<REAL, R_, BWD_SOLVE, W_, false>))
#define SET_N_W(R_) \
    switch (L.ex.waves_per_block) { case 1: SET_N(R_, 1) break; case 2: SET_N(R_, 2) break; case 4: SET_N(R_, 4) break; default: SET_N(R_, 8) break; }
            switch (pack_width) {
                case 64: SET_N_W(1) break;
                case 128: SET_N_W(2) break;
                default: SET_N_W(4) break;
            }
#undef SET_N_W
#undef SET_N
            if (pack_width == 128) {  // the non-temporal instantiations (n12_nt)
                if (L.ex.waves_per_block == 4) { SET_N1((k_fwd_narrow<REAL, 2, FWD_SOLVE, 4, false, true>)) SET_N1((k_bwd_narrow<REAL, 2, BWD_SOLVE, 4, false, true>)) }
                if (L.ex.waves_per_block == 8) { SET_N1((k_fwd_narrow<REAL, 2, FWD_SOLVE, 8, false, true>)) SET_N1((k_bwd_narrow<REAL, 2, BWD_SOLVE, 8, false, true>)) }
            }
#undef SET_N1
        }
        if ((rc = dalloc(&d_tmp0, n_layers))) return rc;
        if ((rc = dalloc(&d_tmp1, n_layers))) return rc;
        if ((rc = dalloc(&d_sol, n_layers))) return rc;
        if ((rc = dalloc(&d_delta_var, 2 * n_vars))) return rc;
        if ((rc = dalloc(&d_delta_c, 2 * n_vars))) return rc;
        if ((rc = dalloc(&d_lb_partial, nb_.n_packs + wb_.n_packs + hb_.n_packs))) return rc;
        HIPCHK(hipHostMalloc((void**)&h_lb, 3 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
        HIPCHK(hipHostGetDevicePointer((void**)&d_lb, h_lb, 0));
        HIPCHK(hipHostMalloc((void**)&h_lb_seq, 3 * sizeof(uint64_t), hipHostMallocMapped | hipHostMallocCoherent));
        HIPCHK(hipHostGetDevicePointer((void**)&d_lb_seq, h_lb_seq, 0));
        std::memset((void*)h_lb_seq, 0, 3 * sizeof(uint64_t));
        if ((rc = dalloc(&d_counts, 4))) return rc;
        HIPCHK(hipMemsetAsync(d_F, 0, n_slots * sizeof(REAL), stream));
        HIPCHK(hipMemsetAsync(d_T, 0, n_slots * sizeof(REAL), stream));
        HIPCHK(hipMemsetAsync(d_lohi, 0, 2 * n_layers * sizeof(REAL), stream));
        HIPCHK(hipMemsetAsync(d_mm_binned, 0, n_layers * sizeof(REAL), stream));  // bdd_cuda_base.cu:45
        for (REAL* p : {d_delta_var, d_delta_c}) HIPCHK(hipMemsetAsync(p, 0, 2 * n_vars * sizeof(REAL), stream));
        for (REAL* p : {d_delta_lay, d_delta_lay_c}) HIPCHK(hipMemsetAsync(p, 0, 2 * n_layers * sizeof(REAL), stream));
#define SET_DYN(K, BYTES) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES)))
        SET_DYN((k_exchange_reduce<REAL, BDDMMA_EX_ACC, EX_ITER>), exch_lds);
        SET_DYN((k_exchange_reduce<REAL, BDDMMA_EX_ACC, EX_RAW>), exch_lds);
        SET_DYN((k_exchange_reduce<REAL, BDDMMA_EX_ACC, EX_ITER, EX_THREADS, EX_UNROLL, EX_NPT, true>), exch_lds);  // run_plain()'s instantiation
        opts_variant = opts ? opts->variant_flags : 0u;
        mixed_fwd = (opts_variant & 2u) == 0;
        // measured in double: 7.1 M nodes (490 MB resident) lose 12 % with non-temporal potentials, 10.5 M (720 MB) gain 4 %
        beyond_cache = dev_bytes > (640ull << 20);   // array bytes (the working set), not arena capacity
        nt_potentials = (sizeof(REAL) == 8 && beyond_cache) ? 1u : 0u;
        exch_small = vars_per_bin <= EXS_MAX_VARS_PER_BIN;  // 256-thread workgroups (kernels.hpp: EXS_*)
        exch_medium = !exch_small && vars_per_bin <= EXM_MAX_VARS_PER_BIN;  // 512-thread workgroups (EXM_*)
        // `deterministic`: the scheduled reduction (kernels.hpp: k_exchange_seg; no atomics, fixed order) where every bin fits its tables
        // and the LDS — runs of <= 32 entries per thread, differences + pairs + counts of a bin in LDS —, else (or with variant_flags
        // bit 17) the per-variable gathers k_delta_gather + k_exchange_bcast.  Both sum in (variable, bdd) order: bit-equal results.
        if (deterministic && !entry_by_var && !big && !(opts_variant & 0x20000u)) {
            const uint32_t T = exch_small ? EXS_THREADS : exch_medium ? EXM_THREADS : EX_THREADS;
            SegExchange SE;
            build_seg_exchange(L, T, (uint32_t)sizeof(REAL), SE);
            constexpr uint32_t VEC = 16 / sizeof(REAL);
            if (SE.ok && SE.max_groups <= 4 && SE.max_entries <= (uint64_t)SEG_MAXL * T * VEC) {
                const uint32_t Z = (SE.max_entries + VEC - 1) / VEC * VEC;
                seg_tile_off = ((Z + 1) * (uint32_t)sizeof(REAL) + 15u) & ~15u;
                seg_cnt_off = seg_tile_off + SE.max_slots * 2u * (uint32_t)sizeof(REAL);
                seg_lds = (seg_cnt_off + SE.max_slots * 2u + 15u) & ~15u;
                if (seg_lds + 1024 <= lds_cu) {
                    if ((rc = upload(&d_seg_bin, SE.bin))) return rc;
                    if ((rc = upload(&d_seg_perm, SE.perm))) return rc;
                    if ((rc = upload(&d_seg_thr, SE.thr))) return rc;
                    use_seg = true;
                    seg_groups = SE.max_groups;
#define SET_SEG_G(T_, G_)                                             \
    SET_DYN((k_exchange_seg<REAL, T_, G_, false>), seg_lds);          \
    SET_DYN((k_exchange_seg<REAL, T_, G_, true>), seg_lds)
#define SET_SEG(T_) if (seg_groups <= 2) { SET_SEG_G(T_, 2); } else { SET_SEG_G(T_, 4); }
                    if (exch_small) { SET_SEG(EXS_THREADS) } else if (exch_medium) { SET_SEG(EXM_THREADS) } else { SET_SEG(EX_THREADS) }
#undef SET_SEG
#undef SET_SEG_G
                }
            }
        }
#undef SET_DYN
        // resident sweeps: chosen when every narrow pack fits its wave's LDS slice and the instance is small enough that the streaming
        // kernels are latency-bound (few waves per SIMD); resident_sweeps = 1 turns them off, = 2 forces them on
        if ((rc = upload(&d_pack_hdr, L.res.pack_hdr, 34))) return rc;
        if ((rc = upload(&d_quad_hdr, L.res.quad_hdr, 35))) return rc;
        res_hdr_ok = L.res.ok && nb_.n_packs > 0;
        if (nb_.n_packs && L.res.ok) {
            res_ns = (L.res.max_slots + 255) / 256 * 256;
            res_nl = (L.res.max_layers + 127) / 128 * 128;
            res_lds = wpb * stage_cap * 2 * (uint32_t)sizeof(REAL) + wpb * res_wave_bytes(sizeof(REAL), res_ns, res_nl);
            const uint32_t static_lds = wpb * (2 * (pack_width + 2) * (uint32_t)sizeof(REAL) + 512) + seg_bytes(wpb);
            const uint32_t mode = opts ? opts->resident_sweeps : 0;
            const bool fits = res_lds + static_lds <= lds_cu - 512;
            // automatic choice: small instances only — fewer than 3 waves per SIMD (2880 packs: with the round-2 kernels the resident sweeps
            // win by 3-6 % from 1 900 to 2 800 packs and lose 10 % at 3 125), all workgroups in flight at once with their LDS slices.  Measured (float, sweep fwd / bwd in us, streaming vs resident): 1 M nodes (1563 packs) 12.1 / 12.2 vs 11.4 / 10.5;
            // 2 M 18.5 / 19.1 vs 20.5 / 19.7; 4 M 27.3 / 25.5 vs 41.8 / 39.0; 10.5 M 50 / 45 vs 108 / 98 — with more waves per SIMD the
            // streaming kernels hide their latency and the resident ones only lose occupancy to their LDS footprint.
            const uint64_t wgs_per_cu = fits ? (lds_cu) / (res_lds + static_lds) : 0;
            const bool all_in_flight = (uint64_t)nb_.n_packs * 256 <= 2880ull * n_cus && (uint64_t)cdiv(nb_.n_packs, wpb) <= (uint64_t)n_cus * std::min<uint64_t>(wgs_per_cu, 2048 / (64 * wpb));
            use_res = fits && mode != 1 && (mode == 2 || all_in_flight) && !big;
            // Second generation (kernels.hpp: k_fwd_res2 / k_bwd_res2), packs of 64 slots with layers of <= 2 nodes: its LDS regions are sized in
            // whole 1 KiB pieces of REAL values (no node words in LDS).  Chosen, like the first, while the packs are (nearly) all in flight at
            // once.  Measured on random set cover (tools/sweep_res2.sh, profiles/r04_res2_sweep.txt), float, it/s streaming -> resident: 1.05 M
            // nodes (1 563 packs) 28.1 k -> 35.0 k, 1.6 M 21.4 k -> 26.1 k, 2.1 M 19.0 k -> 19.9 k, 3.1 M (4 688 packs, 1.4 x what the CUs' LDS holds
            // at once) 16.3 k -> 16.7 k, 4.2 M (128-slot streaming packs) 14.5 k -> 13.4 k; double gains only with every pack in flight.
            // variant_flags bit 11: first generation only.
            if (pack_width == 64 && !narrow_seg && mode != 1 && !big && !(opts && (opts->variant_flags & 0x800u))) {
                res2_ns = res2_slot_capacity((uint32_t)sizeof(REAL), L.res.max_slots);
                res2_nl = res2_layer_capacity((uint32_t)sizeof(REAL), L.res.max_layers);
                res2_lds = wpb * stage_cap * 2 * (uint32_t)sizeof(REAL) + wpb * res2_wave_bytes(sizeof(REAL), res2_ns, res2_nl);
                const uint64_t wgs2 = res2_lds + 256 <= lds_cu ? std::min<uint64_t>((lds_cu) / (res2_lds + 256), 2048 / (64 * wpb)) : 0;
                const uint64_t in_flight = (uint64_t)n_cus * wgs2 * wpb;  // packs the chip holds at once
                const bool auto_ok = sizeof(REAL) == 4 ? (uint64_t)nb_.n_packs * 100 <= in_flight * 145 : (uint64_t)nb_.n_packs <= in_flight;
                if (wgs2 && (mode == 2 || auto_ok)) {
                    Res2Records R2;
                    build_res2_records(L, sizeof(REAL), res2_ns, res2_nl, R2);
                    if (R2.ok && R2.rec.size() * sizeof(uint32_t) < (1ull << 31)) {
                        if ((rc = upload(&d_res2_rec, R2.rec))) return rc;
                        if ((rc = upload(&d_res2_rec_off, R2.rec_off))) return rc;
                        res2_n_words = (uint32_t)R2.rec.size();
                        use_res2 = use_res = true;
#define SET_RES2(W_)                                                                                                                            \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fwd_res2<REAL, W_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)res2_lds)); \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bwd_res2<REAL, W_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)res2_lds));
                        switch (wpb) { case 1: SET_RES2(1) break; case 2: SET_RES2(2) break; case 4: SET_RES2(4) break; default: SET_RES2(8) break; }
#undef SET_RES2
                    }
                }
            }
            if (use_res && !use_res2) {
#define SET_RES(R_, W_) \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fwd_res<REAL, R_, W_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(res_lds + seg_bytes(wpb)))); \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bwd_res<REAL, R_, W_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(res_lds + seg_bytes(wpb))));
#define SET_RES_W(R_) \
    switch (wpb) { case 1: SET_RES(R_, 1) break; case 2: SET_RES(R_, 2) break; case 4: SET_RES(R_, 4) break; default: SET_RES(R_, 8) break; }
                switch (pack_width) {
                    case 64: SET_RES_W(1) break;
                    case 128: SET_RES_W(2) break;
                    default: SET_RES_W(4) break;
                }
#undef SET_RES_W
#undef SET_RES
            }
        }
        // streaming solve sweeps, second generation (kernels.hpp: k_fwd_narrow2 / k_bwd_narrow2): every narrow pack; the general form (GEN) where
        // layers are wider than two nodes or packs are staggered.  variant_flags bit 12: first generation; bit 13: records also where they are not
        // shared (the differential tests run the general form on random instances that way)
        uint32_t narrow2_static = 0;
        if (nb_.n_packs && !use_res && !wb_.n_packs && !(opts && (opts->variant_flags & 0x1000u))) {
            StreamRecords SR;
            build_stream_records(L, sizeof(REAL), SR);
            narrow2_static = L.ex.waves_per_block * ((2 * pack_width + 2 + pack_width + 2) * (uint32_t)sizeof(REAL) + 3 * 64 * 4 + 64);
            // ... where the records are shared by the packs of a structure template (every row of a constraint family): 16 bytes per lane and hop
            // that are not shared cost more bandwidth than the instructions they save (general linear rows: k_fwd_mixed).  Limit: the records
            // hold at most a quarter of what the potentials of the narrow packs do.
            // ... and, in float, while the instance is small enough for the caches to help (tools/kbench.py, same-box A/B first / second
            // generation, it/s): 10.5 M nodes 8 190 / 8 540, 21 M 3 520 / 3 495, 42 M 1 605 / 1 596, 105 M 642 / 618 — once the sweeps run at
            // the HBM rate the instructions saved buy nothing and the records' L2 traffic costs a little; double: 4 273 / 4 353 and 339 / 357.
            const bool forced = opts && (opts->variant_flags & 0x2000u);
            const bool shared = forced || (SR.rec.size() * sizeof(uint32_t) <= (uint64_t)L.narrow_slots * sizeof(REAL) / 2 &&
                                           (sizeof(REAL) == 8 || n_slots <= 16'000'000ull));
            if (SR.ok && shared && stage_lds + narrow2_static + seg_bytes(L.ex.waves_per_block) <= 64 * 1024) {
                if ((rc = upload(&d_srec, SR.rec))) return rc;
                if ((rc = upload(&d_srec_off, SR.rec_off))) return rc;
                srec_words = (uint32_t)SR.rec.size();
                use_narrow2 = true;
                narrow_gen = narrow_seg;
                for (uint16_t r : L.narrow.hop_root) narrow_gen = narrow_gen || r != NO_ROOT;
            }
        }
        // streaming solve sweeps, third generation (kernels/narrow3.hpp): a lane per LAYER — packs of 128 slots whose layers have <= 2 nodes and whose
        // hops have <= 64 layers, no staggered packs, records shared like the second generation's.  At every size: what it saves — issue slots and bytes through the CU's vector-memory path — is what
        // bounds the sweeps with or without the caches' help (profiles/r05_hbm_only.txt).  variant_flags bit 18: second / first generation instead.
        if (nb_.n_packs && !use_res && !wb_.n_packs && !hb_.n_packs && !narrow_seg && pack_width == 128 && (L.ex.waves_per_block == 4 || L.ex.waves_per_block == 8) &&
            !(opts && (opts->variant_flags & 0x41000u))) {
            LayerRecords LR;
            build_layer_records(L, sizeof(REAL), LR);
            const uint32_t n3_static = L.ex.waves_per_block * (4 * (128 + 128) * (uint32_t)sizeof(REAL) + 3 * 64 * 4);
            const bool forced = opts && (opts->variant_flags & 0x2000u);
            // ... and while the instance is small enough for the Infinity Cache to matter (same-box A/B against what the rules chose before,
            // tools/ab_n3.sh, it/s float / double: 4.2 M nodes 16 150 / 11 660 against 15 380 / 10 770; 10.5 M 8 850 / 4 390 against 8 530 /
            // 4 290; 21 M 3 480 / 1 838 against 3 555 / 1 879; 42 M 1 604 / 888 against 1 638 / 913): beyond, a shorter hop loop only
            // lengthens the staging phase of the same waves (profiles/r05_hbm_only.txt) and the eight-pack first / second generation stay ahead
            const bool shared = forced || (LR.rec.size() * sizeof(uint32_t) <= (uint64_t)L.narrow_slots * sizeof(REAL) / 2 && n_slots <= 16'000'000ull);
            if (LR.ok && shared && stage_lds + n3_static <= lds_cu - 1024) {
                if ((rc = upload(&d_lrec, LR.rec))) return rc;
                if ((rc = upload(&d_lrec_off, LR.rec_off))) return rc;
                lrec_words = (uint32_t)LR.rec.size();
                use_narrow3 = true;
                // double, footprint beyond the Infinity Cache's reach (the condition of the non-temporal stores): the instantiation that LOADS what a
                // sweep reads once — potentials, staging tables — non-temporally (profiles/r05_placement.txt #7: 10.5 M nodes +3.5 %)
                n3_nt = sizeof(REAL) == 8 && nt_potentials != 0;
#define SET_N3(W_, NT_)                                                                                                                              \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fwd_narrow3<REAL, W_, NT_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage_lds)); \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bwd_narrow3<REAL, W_, NT_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage_lds));
                if (L.ex.waves_per_block == 4) { SET_N3(4, false) } else { SET_N3(8, false) }
                if constexpr (sizeof(REAL) == 8) {
                    if (L.ex.waves_per_block == 4) { SET_N3(4, true) } else { SET_N3(8, true) }
                }
#undef SET_N3
            }
        }
        // First / second (double only) generation where the third stops — beyond 16 M slots, i.e. beyond the Infinity Cache's reach: the same
        // policy, the instantiation that loads potentials and staging tables non-temporally (round 5's build knobs BDDMMA_LD_POT_AUX / _TAB_AUX as a
        // template parameter; profiles/r06_nt_loads.txt, alternating with the library before, it/s: float 21 M nodes 3 705 / 3 837 -> 3 823 / 4 006,
        // 42 M 1 689 / 1 808 -> 1 836 / 1 891, 105 M 669 / 652 -> 671 / 661; double 21 M 1 938 / 2 090 -> 1 996 / 2 150, 42 M and 105 M +0-1 %).
        // Packs of 128 slots, 4 / 8 per workgroup, layers of at most two nodes, no staggered packs: what instances of that size have (the other
        // instantiations are not built).  variant_flags bit 20: whatever the footprint (the tests' way to these kernels).
        if (nb_.n_packs && !use_res && !use_narrow3 && !wb_.n_packs && !hb_.n_packs && !narrow_seg && !narrow_gen && pack_width == 128 &&
            (L.ex.waves_per_block == 4 || L.ex.waves_per_block == 8) && (beyond_cache || (opts && (opts->variant_flags & 0x100000u))))
#ifndef BDDMMA_EXP_NO_N12_NT   // A/B builds (tools/build_variant.sh)
            n12_nt = !use_narrow2 || sizeof(REAL) == 8;
#else
            n12_nt = false;
#endif
        nt_loads = n3_nt || n12_nt;
        if (wb_.n_packs) {
            // one workgroup per wide pack, thread t owns the nodes t + i * wide_threads of a hop (kernels.hpp: k_fwd_wide2)
            // two nodes of a hop per thread: half the wavefronts at the hop's two barriers and two independent chains per lane.  Wide-only
            // instance (25 000 rows of 18 variables, 15 M nodes): packs of 512 / 1024 slots with one node per thread 1 823 / 1 662 it/s, with
            // two 2 158 / 2 205, with four (1024 slots) 1 936
            wide_threads = std::min<uint32_t>(1024, std::max<uint32_t>(64, ((wide_pack_width + 1) / 2 + 63) / 64 * 64));
            wide_npt = (wide_pack_width + wide_threads - 1) / wide_threads;
            wide_npt = wide_npt <= 1 ? 1 : (wide_npt <= 2 ? 2 : 4);
            wide_lds = (uint32_t)wide2_lds_bytes(sizeof(REAL), wide_pack_width, true);
            if (wide_lds > lds_cu || wide_pack_width > 4096) {
                err = "wide_pack_width " + std::to_string(wide_pack_width) + " needs " + std::to_string(wide_lds) + " B of LDS (> 160 KiB)";
                return BDDMMA_ERR_UNSUPPORTED;
            }
#define SET_LDS(K) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wide_lds))
#define SET_LDS_N(N_) \
    SET_LDS((k_fwd_wide2<REAL, FWD_PLAIN, N_>)); SET_LDS((k_fwd_wide2<REAL, FWD_SOLVE, N_>)); SET_LDS((k_fwd_wide2<REAL, FWD_SOLUTION, N_>)); \
    SET_LDS((k_bwd_wide2<REAL, BWD_PLAIN, N_>)); SET_LDS((k_bwd_wide2<REAL, BWD_SOLVE, N_>)); SET_LDS((k_bwd_wide2<REAL, BWD_MARGINALS, N_>));
            if (wide_npt == 1) { SET_LDS_N(1) } else if (wide_npt == 2) { SET_LDS_N(2) } else { SET_LDS_N(4) }
#undef SET_LDS_N
#undef SET_LDS
        }
        // one launch for narrow + wide solve sweeps when the wide packs fit the narrow workgroup size (<= 4 nodes per thread) and their
        // LDS frontier does not cost the narrow workgroups their occupancy
        if (wb_.n_packs && nb_.n_packs && !use_res && !(opts && (opts->variant_flags & 1u))) {
            const uint32_t threads = 64 * wpb;
            const uint32_t npt = (wide_pack_width + threads - 1) / threads;
            const uint32_t narrow_dyn = stage_lds + seg_bytes(wpb);
            // (no launch attribute is set for the mixed kernels: static + dynamic LDS has to stay within the default 64 KiB)
            if (npt <= 4 && wide_lds <= narrow_dyn + 16 * 1024 && std::max(narrow_dyn, wide_lds) + narrow_static <= 64 * 1024) {
                mixed = true;
                mixed_npt = npt <= 1 ? 1 : (npt <= 2 ? 2 : 4);
                mixed_lds = std::max(narrow_dyn, wide_lds);
            }
        }
        // Whole iterations in one launch (kernels/small.hpp) where every pack fits one workgroup: the second-generation resident records (their
        // conditions: 64-slot packs, layers of <= 2 nodes, one stage group per pack and one staging round per quad), narrow packs only, the LDS-atomic
        // exchange's arithmetic (not the deterministic / by-variable orders), all of it within the CU's LDS.  variant_flags bit 19: off.
        if (use_res2 && res_hdr_ok && nb_.n_packs <= SMALL_MAX_PACKS && !wb_.n_packs && !hb_.n_packs && !deterministic && !entry_by_var && !big &&
            L.var_ptr.size() == n_vars + 1 && L.ex.vpos.size() == n_layers && !(opts && (opts->variant_flags & 0x80000u))) {
            const uint32_t np = nb_.n_packs;
            std::vector<uint32_t> lds_off(n_layers);
            uint32_t max_hops = 0;
            bool ok = true;
            for (uint32_t k = 0; k < (uint32_t)n_layers && ok; ++k) {
                const uint32_t l = L.var_layers[k];
                uint32_t p = 0;   // the pack whose layer range holds l (pack_hdr: [2] first layer, [3] layers)
                while (p < np && !(l >= L.res.pack_hdr[8 * (size_t)p + 2] && l < L.res.pack_hdr[8 * (size_t)p + 2] + L.res.pack_hdr[8 * (size_t)p + 3])) ++p;
                if (p == np || l - L.res.pack_hdr[8 * (size_t)p + 2] >= res2_nl) { ok = false; break; }
                lds_off[k] = (p * small_stage_stride(res2_nl) + (l - L.res.pack_hdr[8 * (size_t)p + 2])) * 2u * (uint32_t)sizeof(REAL);
            }
            for (uint32_t p = 0; p < np; ++p) max_hops = std::max(max_hops, L.res.pack_hdr[8 * (size_t)p + 5] & 0xFFFFu);
            small = SmallDev{};
            small.ns = res2_ns; small.nl = res2_nl; small.n_packs = np; small.wpb = wpb; small.n_quads = cdiv(np, wpb);
            small.n_vars = (uint32_t)n_vars; small.n_entries = (uint32_t)n_layers; small.rec_words = res2_n_words;
            small.rec_cap = (max_hops + 1u) * 1024u;   // + the spare KiB the forward sweep's one-hop-ahead read lands in behind the last hop
            uint32_t bytes = small_lds_bytes((uint32_t)sizeof(REAL), np, (uint32_t)n_vars, (uint32_t)n_layers, small);
            small_rl = bytes + 512 <= lds_cu;
            if (!small_rl) {
                small.rec_cap = 0;
                bytes = small_lds_bytes((uint32_t)sizeof(REAL), np, (uint32_t)n_vars, (uint32_t)n_layers, small);
            }
            if (ok && bytes + 512 <= lds_cu) {
                if ((rc = upload(&d_small_lds, lds_off))) return rc;
                small.pack_hdr = d_pack_hdr; small.quad_hdr = d_quad_hdr; small.rec = d_res2_rec; small.rec_off = d_res2_rec_off;
                small.var_ptr = d_var_ptr; small.var_lds = d_small_lds; small.var_ent = d_vpos;
                small_lds = bytes;
                small_nw = np <= 1 ? 1 : np <= 2 ? 2 : np <= 4 ? 4 : np <= 8 ? 8 : 16;
#define SET_SMALL(NW_)                                                                                                                                  \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_iterate_small<REAL, NW_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds)); \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_iterate_small<REAL, NW_, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds));
                switch (small_nw) { case 1: SET_SMALL(1) break; case 2: SET_SMALL(2) break; case 4: SET_SMALL(4) break; case 8: SET_SMALL(8) break; default: SET_SMALL(16) break; }
#undef SET_SMALL
                small_ok = true;
                fused_small = true;
            }
        }
        solve_sweep_kind = !nb_.n_packs ? BDDMMA_SWEEPS_NONE : mixed ? BDDMMA_SWEEPS_MIXED : (use_res && use_res2) ? BDDMMA_SWEEPS_RESIDENT2 : use_res ? BDDMMA_SWEEPS_RESIDENT1
                           : use_narrow3 ? BDDMMA_SWEEPS_STREAMING3 : use_narrow2 ? BDDMMA_SWEEPS_STREAMING2 : BDDMMA_SWEEPS_STREAMING1;
        HIPCHK(hipStreamSynchronize(stream));
        return BDDMMA_OK;
    }

    // ---------------------------------------------------------------------------- launches
    DevPtrs<REAL> ptrs(const REAL* delta_lay) const
    {
        DevPtrs<REAL> d;
        d.nwords = d_nwords; d.n_nwords = n_nwords; d.wwords = d_wwords; d.wide_slot_base = wide_slot_base;
        d.F = d_F; d.T = d_T; d.lohi = d_lohi;
        d.delta_lay = delta_lay; d.mm_binned = d_mm_binned; d.lpos = d_lpos; d.cs_entry = d_cs_entry; d.cs_slot = d_cs_slot;
        d.n_slots = (uint32_t)n_slots; d.n_layers = (uint32_t)n_layers; d.n_narrow_layers = n_narrow_layers;
        d.lb_partial = d_lb_partial;
        d.mm0_out = d_tmp0; d.mm1_out = d_tmp1; d.sol_out = d_sol;
        d.x_layer = d_x_layer;  // nullptr until an L-BFGS wrapper asks for it (lbfgs_views)
        d.stop = run_stop;
        d.run_iter = run_iter;
        d.big = big ? 1u : 0u;
        return d;
    }
    PackDev pdev(const PackBufs& b, uint32_t lb_base, uint32_t seg_off = 0) const
    {
        const bool narrow_set = &b == &nb_;
        return PackDev{b.pack_hop_ptr, b.hop_node_off, b.hop_layer_off, b.pack_steps, b.hop_root, d_pack_word_off,
                       d_pack_group_ptr, d_grp_layer_off, d_grp_hop_end, d_quad_round_ptr, d_cs_ptr, stage_cap, seg_off, b.n_packs, lb_base,
                       nt_potentials, xcd_chunk(), narrow_set ? n2_hdr_pack() : nullptr, narrow_set ? n2_hdr_quad() : nullptr};
    }
    // dynamic LDS of a narrow launch with `w` waves per workgroup: `base` bytes of the kernel's own use, then (only when some pack has
    // layers wider than two nodes) the seg_min2 scratch, 128 REALs per wave
    uint32_t seg_bytes(uint32_t w) const { return narrow_seg ? w * 128 * (uint32_t)sizeof(REAL) : 0; }
    // block -> pack map of the narrow launches (kernels.hpp: block_to_pack): chunks of 32 workgroups interleaved over the XCDs
    // — chosen when contiguous eighths would be unbalanced (init: hop counts per eighth differ by more than 10 %: instances that mix
    // constraint families); homogeneous instances keep the contiguous map, which measured 1-3 % faster there (neighbouring packs share
    // more L2 lines).  variant_flags bit 7 forces the contiguous map, bit 8 the interleaved one.
    uint32_t xcd_chunk_auto = 0;
    uint32_t xcd_chunk() const { return xcd_chunk_auto; }
    uint32_t narrow_grid(uint32_t n_quads) const
    {
        const uint32_t c = xcd_chunk();
        return c ? 8 * c * cdiv(n_quads, 8 * c) : 8 * cdiv(n_quads, 8);
    }

    template <int MODE>
    int launch_fwd(const REAL* delta_lay, REAL omega, int kclass)
    {
        DevPtrs<REAL> d = ptrs(delta_lay);
        prof_begin(kclass);
        hipStream_t sw = stream;
        if (mixed && mixed_fwd && MODE == FWD_SOLVE) {
            const PackDev pkw = pdev(wb_, nb_.n_packs), pkn = pdev(nb_, 0, stage_lds);
            const uint32_t nw8 = (wb_.n_packs + 7u) & ~7u;
            const dim3 grid(nw8 + narrow_grid(cdiv(nb_.n_packs, wpb))), block(64 * wpb);
#define LAUNCH_M(R_, W_)                                                                                                                                   \
    switch (mixed_npt) {                                                                                                                                   \
        case 1: hipLaunchKernelGGL((k_fwd_mixed<REAL, R_, W_, 1>), grid, block, mixed_lds, stream, d, pkn, pkw, omega, wide_pack_width); break;            \
        case 2: hipLaunchKernelGGL((k_fwd_mixed<REAL, R_, W_, 2>), grid, block, mixed_lds, stream, d, pkn, pkw, omega, wide_pack_width); break;            \
        default: hipLaunchKernelGGL((k_fwd_mixed<REAL, R_, W_, 4>), grid, block, mixed_lds, stream, d, pkn, pkw, omega, wide_pack_width); break;           \
    }
#define LAUNCH_MW(R_) \
    switch (wpb) { case 1: LAUNCH_M(R_, 1) break; case 2: LAUNCH_M(R_, 2) break; case 4: LAUNCH_M(R_, 4) break; default: LAUNCH_M(R_, 8) break; }
            switch (pack_width) {
                case 64: LAUNCH_MW(1) break;
                case 128: LAUNCH_MW(2) break;
                default: LAUNCH_MW(4) break;
            }
#undef LAUNCH_MW
#undef LAUNCH_M
        } else {
        if (wb_.n_packs) {
            const PackDev pk = pdev(wb_, nb_.n_packs);
            const dim3 g(wb_.n_packs), b(wide_threads);
            switch (wide_npt) {
                case 1: hipLaunchKernelGGL((k_fwd_wide2<REAL, MODE, 1>), g, b, wide_lds, sw, d, pk, omega, wide_pack_width); break;
                case 2: hipLaunchKernelGGL((k_fwd_wide2<REAL, MODE, 2>), g, b, wide_lds, sw, d, pk, omega, wide_pack_width); break;
                default: hipLaunchKernelGGL((k_fwd_wide2<REAL, MODE, 4>), g, b, wide_lds, sw, d, pk, omega, wide_pack_width); break;
            }
        }
        if (nb_.n_packs) {
            // SOLVE sweeps: `wpb` packs per workgroup with cooperative staging; the other modes stage nothing
            const uint32_t w = (MODE == FWD_SOLVE) ? wpb : 1;
            const bool res = use_res && MODE == FWD_SOLVE;
            const bool two_node = MODE == FWD_SOLVE && !narrow_seg;  // SEG = false instantiation (kernels.hpp); other modes: <..., true> again
            const uint32_t base_lds = res ? res_lds : ((MODE == FWD_SOLVE) ? stage_lds : 0);
            const uint32_t dyn = base_lds + seg_bytes(w);
            const PackDev pk = pdev(nb_, 0, base_lds);
            const dim3 grid(narrow_grid(cdiv(nb_.n_packs, w))), block(64 * w);
            const ResDev rd{d_pack_hdr, d_quad_hdr, res_ns, res_nl};
#define LAUNCH_N(R_, W_)                                                                                                      \
    if (res && use_res2) hipLaunchKernelGGL((k_fwd_res2<REAL, W_>), grid, block, res2_lds, stream, rd.pack_hdr, rd.quad_hdr, res2_ns, res2_nl, pk.n_packs, pk.xcd_chunk, d.stop, d.run_iter, d_res2_rec, d_res2_rec_off, res2_n_words, d, pk, omega); \
    else if (res) hipLaunchKernelGGL((k_fwd_res<REAL, R_, W_>), grid, block, dyn, stream, rd.pack_hdr, rd.quad_hdr, rd.ns, rd.nl, pk.n_packs, pk.xcd_chunk, d.stop, d.run_iter, d, pk, omega);                           \
    else if (MODE == FWD_SOLVE && use_narrow3 && n3_nt && (W_ == 4 || W_ == 8)) hipLaunchKernelGGL((k_fwd_narrow3<REAL, (W_ == 8 ? 8 : 4), sizeof(REAL) == 8>), grid, block, dyn, stream, d, pk, d_lrec, d_lrec_off, lrec_words, omega); \
    else if (MODE == FWD_SOLVE && use_narrow3 && (W_ == 4 || W_ == 8)) hipLaunchKernelGGL((k_fwd_narrow3<REAL, (W_ == 8 ? 8 : 4)>), grid, block, dyn, stream, d, pk, d_lrec, d_lrec_off, lrec_words, omega); \
    else if (MODE == FWD_SOLVE && use_narrow2 && n12_nt && R_ == 2 && (W_ == 4 || W_ == 8)) hipLaunchKernelGGL((k_fwd_narrow2<REAL, 2, (W_ == 8 ? 8 : 4), false, sizeof(REAL) == 8>), grid, block, dyn, stream, d, pk, d_srec, d_srec_off, srec_words, omega); \
    else if (MODE == FWD_SOLVE && !use_narrow2 && n12_nt && R_ == 2 && (W_ == 4 || W_ == 8)) hipLaunchKernelGGL((k_fwd_narrow<REAL, 2, FWD_SOLVE, (W_ == 8 ? 8 : 4), false, true>), grid, block, dyn, stream, d, pk, omega); \
    else if (MODE == FWD_SOLVE && use_narrow2 && narrow_gen) hipLaunchKernelGGL((k_fwd_narrow2<REAL, R_, W_, true>), grid, block, dyn, stream, d, pk, d_srec, d_srec_off, srec_words, omega); \
    else if (MODE == FWD_SOLVE && use_narrow2) hipLaunchKernelGGL((k_fwd_narrow2<REAL, R_, W_, false>), grid, block, dyn, stream, d, pk, d_srec, d_srec_off, srec_words, omega); \
    else if (two_node) hipLaunchKernelGGL((k_fwd_narrow<REAL, R_, MODE, W_, MODE != FWD_SOLVE>), grid, block, dyn, stream, d, pk, omega); \
    else hipLaunchKernelGGL((k_fwd_narrow<REAL, R_, MODE, W_>), grid, block, dyn, stream, d, pk, omega)
#define LAUNCH_W(R_) \
    switch (w) { case 1: LAUNCH_N(R_, 1); break; case 2: LAUNCH_N(R_, 2); break; case 4: LAUNCH_N(R_, 4); break; default: LAUNCH_N(R_, 8); break; }
            switch (pack_width) {
                case 64: LAUNCH_W(1); break;
                case 128: LAUNCH_W(2); break;
                default: LAUNCH_W(4); break;
            }
#undef LAUNCH_W
#undef LAUNCH_N
        }
        }  // !mixed
        if (hb_.n_packs) {
            const PackDev pk = pdev(hb_, nb_.n_packs + wb_.n_packs);
            hipLaunchKernelGGL((k_fwd_wide<REAL, MODE, true>), dim3(hb_.n_packs), dim3(WIDE_THREADS), 0, stream, d, pk, omega, huge_pack_width, d_huge_scratch);
        }
        prof_end(kclass);
        HIPCHK(hipGetLastError());
        return BDDMMA_OK;
    }
    template <int MODE>
    int launch_bwd(const REAL* delta_lay, REAL omega, int kclass)
    {
        lb_cached = false;
        ++lb_gen;
        DevPtrs<REAL> d = ptrs(delta_lay);
        prof_begin(kclass);
        hipStream_t sw = stream;
        if (mixed && MODE == BWD_SOLVE) {
            const PackDev pkw = pdev(wb_, nb_.n_packs), pkn = pdev(nb_, 0, stage_lds);
            const uint32_t nw8 = (wb_.n_packs + 7u) & ~7u;
            const dim3 grid(nw8 + narrow_grid(cdiv(nb_.n_packs, wpb))), block(64 * wpb);
#define LAUNCH_M(R_, W_)                                                                                                                                   \
    switch (mixed_npt) {                                                                                                                                   \
        case 1: hipLaunchKernelGGL((k_bwd_mixed<REAL, R_, W_, 1>), grid, block, mixed_lds, stream, d, pkn, pkw, omega, wide_pack_width); break;            \
        case 2: hipLaunchKernelGGL((k_bwd_mixed<REAL, R_, W_, 2>), grid, block, mixed_lds, stream, d, pkn, pkw, omega, wide_pack_width); break;            \
        default: hipLaunchKernelGGL((k_bwd_mixed<REAL, R_, W_, 4>), grid, block, mixed_lds, stream, d, pkn, pkw, omega, wide_pack_width); break;           \
    }
#define LAUNCH_MW(R_) \
    switch (wpb) { case 1: LAUNCH_M(R_, 1) break; case 2: LAUNCH_M(R_, 2) break; case 4: LAUNCH_M(R_, 4) break; default: LAUNCH_M(R_, 8) break; }
            switch (pack_width) {
                case 64: LAUNCH_MW(1) break;
                case 128: LAUNCH_MW(2) break;
                default: LAUNCH_MW(4) break;
            }
#undef LAUNCH_MW
#undef LAUNCH_M
        } else {
        if (wb_.n_packs) {
            const PackDev pk = pdev(wb_, nb_.n_packs);
            const dim3 g(wb_.n_packs), b(wide_threads);
            switch (wide_npt) {
                case 1: hipLaunchKernelGGL((k_bwd_wide2<REAL, MODE, 1>), g, b, wide_lds, sw, d, pk, omega, wide_pack_width); break;
                case 2: hipLaunchKernelGGL((k_bwd_wide2<REAL, MODE, 2>), g, b, wide_lds, sw, d, pk, omega, wide_pack_width); break;
                default: hipLaunchKernelGGL((k_bwd_wide2<REAL, MODE, 4>), g, b, wide_lds, sw, d, pk, omega, wide_pack_width); break;
            }
        }
        if (nb_.n_packs) {
            // SOLVE sweeps: `wpb` packs per workgroup with cooperative staging; the other modes stage nothing
            const uint32_t w = (MODE == BWD_SOLVE) ? wpb : 1;
            const bool res = use_res && MODE == BWD_SOLVE;
            const bool two_node = MODE == BWD_SOLVE && !narrow_seg;
            const uint32_t base_lds = res ? res_lds : ((MODE == BWD_SOLVE) ? stage_lds : 0);
            const uint32_t dyn = base_lds + seg_bytes(w);
            const PackDev pk = pdev(nb_, 0, base_lds);
            const dim3 grid(narrow_grid(cdiv(nb_.n_packs, w))), block(64 * w);
            const ResDev rd{d_pack_hdr, d_quad_hdr, res_ns, res_nl};
#define LAUNCH_N(R_, W_)                                                                                                      \
    if (res && use_res2) hipLaunchKernelGGL((k_bwd_res2<REAL, W_>), grid, block, res2_lds, stream, rd.pack_hdr, rd.quad_hdr, res2_ns, res2_nl, pk.n_packs, pk.xcd_chunk, d.stop, d.run_iter, d_res2_rec, d_res2_rec_off, res2_n_words, d, pk, omega); \
    else if (res) hipLaunchKernelGGL((k_bwd_res<REAL, R_, W_>), grid, block, dyn, stream, rd.pack_hdr, rd.quad_hdr, rd.ns, rd.nl, pk.n_packs, pk.xcd_chunk, d.stop, d.run_iter, d, pk, omega);                           \
    else if (MODE == BWD_SOLVE && use_narrow3 && n3_nt && (W_ == 4 || W_ == 8)) hipLaunchKernelGGL((k_bwd_narrow3<REAL, (W_ == 8 ? 8 : 4), sizeof(REAL) == 8>), grid, block, dyn, stream, d, pk, d_lrec, d_lrec_off, lrec_words, omega); \
    else if (MODE == BWD_SOLVE && use_narrow3 && (W_ == 4 || W_ == 8)) hipLaunchKernelGGL((k_bwd_narrow3<REAL, (W_ == 8 ? 8 : 4)>), grid, block, dyn, stream, d, pk, d_lrec, d_lrec_off, lrec_words, omega); \
    else if (MODE == BWD_SOLVE && use_narrow2 && n12_nt && R_ == 2 && (W_ == 4 || W_ == 8)) hipLaunchKernelGGL((k_bwd_narrow2<REAL, 2, (W_ == 8 ? 8 : 4), false, sizeof(REAL) == 8>), grid, block, dyn, stream, d, pk, d_srec, d_srec_off, srec_words, omega); \
    else if (MODE == BWD_SOLVE && !use_narrow2 && n12_nt && R_ == 2 && (W_ == 4 || W_ == 8)) hipLaunchKernelGGL((k_bwd_narrow<REAL, 2, BWD_SOLVE, (W_ == 8 ? 8 : 4), false, true>), grid, block, dyn, stream, d, pk, omega); \
    else if (MODE == BWD_SOLVE && use_narrow2 && narrow_gen) hipLaunchKernelGGL((k_bwd_narrow2<REAL, R_, W_, true>), grid, block, dyn, stream, d, pk, d_srec, d_srec_off, srec_words, omega); \
    else if (MODE == BWD_SOLVE && use_narrow2) hipLaunchKernelGGL((k_bwd_narrow2<REAL, R_, W_, false>), grid, block, dyn, stream, d, pk, d_srec, d_srec_off, srec_words, omega); \
    else if (two_node) hipLaunchKernelGGL((k_bwd_narrow<REAL, R_, MODE, W_, MODE != BWD_SOLVE>), grid, block, dyn, stream, d, pk, omega); \
    else hipLaunchKernelGGL((k_bwd_narrow<REAL, R_, MODE, W_>), grid, block, dyn, stream, d, pk, omega)
#define LAUNCH_W(R_) \
    switch (w) { case 1: LAUNCH_N(R_, 1); break; case 2: LAUNCH_N(R_, 2); break; case 4: LAUNCH_N(R_, 4); break; default: LAUNCH_N(R_, 8); break; }
            switch (pack_width) {
                case 64: LAUNCH_W(1); break;
                case 128: LAUNCH_W(2); break;
                default: LAUNCH_W(4); break;
            }
#undef LAUNCH_W
#undef LAUNCH_N
        }
        }  // !mixed
        if (hb_.n_packs) {
            const PackDev pk = pdev(hb_, nb_.n_packs + wb_.n_packs);
            hipLaunchKernelGGL((k_bwd_wide<REAL, MODE, true>), dim3(hb_.n_packs), dim3(WIDE_THREADS), 0, stream, d, pk, omega, huge_pack_width, d_huge_scratch);
        }
        prof_end(kclass);
        HIPCHK(hipGetLastError());
        return BDDMMA_OK;
    }
    // compute_delta + normalize_delta + broadcast to the layers, see kernels.hpp (k_exchange_reduce)
    void launch_bcast(const REAL* delta_var, REAL* delta_lay, RunGate g = RunGate{}, RunStep run = RunStep{})
    {
        hipLaunchKernelGGL((k_exchange_bcast<REAL>), dim3(cdiv(cdiv(n_layers, 4), 256)), dim3(256), 0, stream, delta_var, d_evar, delta_lay,
                           (uint32_t)n_layers, (uint32_t)n_vars, g, run);
    }
    // un-normalised per-variable sums of the deferred min-marginal differences (compute_delta only)
    void launch_reduce_raw(REAL* delta_var)
    {
        if (deterministic)
            hipLaunchKernelGGL((k_delta_gather<REAL, false>), dim3(cdiv(n_vars, 256)), dim3(256), 0, stream, d_mm_binned, d_var_ptr,
                               d_vpos, delta_var, (uint32_t)n_vars, RunGate{});
        else
            hipLaunchKernelGGL((k_exchange_reduce<REAL, BDDMMA_EX_ACC, EX_RAW>), dim3(n_bins), dim3(EX_THREADS), exch_lds, stream, d_mm_binned,
                               d_bin_ptr, d_bvar, (const uint32_t*)nullptr, 0u, vars_per_bin, (uint32_t)n_vars, (uint32_t)n_layers, d_nbdds, delta_var,
                               (REAL*)nullptr, RunStep{});
    }
    int exchange(bool ends_iteration = false)
    {
        const RunStep rstep = (ends_iteration && run_stop) ? run_step : RunStep{};
        prof_begin(BDDMMA_K_FINISH_DELTA);
        if (entry_by_var) {
            // entries of a variable are consecutive: one thread per variable reduces, normalises and broadcasts (deterministic order)
            hipLaunchKernelGGL((k_exchange_byvar<REAL>), dim3(cdiv(n_vars, 256)), dim3(256), 0, stream, d_mm_binned, d_var_ptr, d_delta_lay,
                               (uint32_t)n_vars, (uint32_t)n_layers, gate(), rstep);
            delta_var_valid = false;
        } else if (deterministic && !use_seg) {
            hipLaunchKernelGGL((k_delta_gather<REAL, true>), dim3(cdiv(n_vars, 256)), dim3(256), 0, stream, d_mm_binned, d_var_ptr,
                               d_vpos, d_delta_var, (uint32_t)n_vars, gate());
            launch_bcast(d_delta_var, d_delta_lay, gate(), rstep);
            delta_var_valid = true;
        } else if (use_seg) {
#define LAUNCH_SEG(T_, RUN_)                                                                                                                        \
    if (seg_groups <= 2) { LAUNCH_SEG_G(T_, 2, RUN_); } else { LAUNCH_SEG_G(T_, 4, RUN_); }
#define LAUNCH_SEG_G(T_, G_, RUN_)                                                                                                                  \
    hipLaunchKernelGGL((k_exchange_seg<REAL, T_, G_, RUN_>), dim3(n_bins + ((RUN_) && rstep.ctl != nullptr ? 1u : 0u)), dim3(T_), seg_lds, stream, d_mm_binned, \
                       reinterpret_cast<const uint4*>(d_seg_bin), gate().stop, gate().iter, reinterpret_cast<const uint4*>(d_seg_perm),              \
                       reinterpret_cast<const uint2*>(d_seg_thr), seg_tile_off, seg_cnt_off, d_delta_lay, rstep)
#define LAUNCH_SEG_R(T_) do { if (run_stop) { LAUNCH_SEG(T_, true) } else { LAUNCH_SEG(T_, false) } } while (0)
            if (exch_small) LAUNCH_SEG_R(EXS_THREADS);
            else if (exch_medium) LAUNCH_SEG_R(EXM_THREADS);
            else LAUNCH_SEG_R(EX_THREADS);
#undef LAUNCH_SEG_R
#undef LAUNCH_SEG
#undef LAUNCH_SEG_G
            delta_var_valid = false;
        } else {
#define LAUNCH_EX(T_, U_, N_, RUN_)                                                                                                                  \
    hipLaunchKernelGGL((k_exchange_reduce<REAL, BDDMMA_EX_ACC, EX_ITER, T_, U_, N_, RUN_>), dim3(n_bins + ((RUN_) && rstep.ctl != nullptr ? 1u : 0u)), dim3(T_), exch_lds, stream, d_mm_binned, d_bin_ptr, \
                       d_bvar, gate().stop, gate().iter, vars_per_bin, (uint32_t)n_vars, (uint32_t)n_layers, d_nbdds, (REAL*)nullptr, d_delta_lay, rstep)
#define LAUNCH_EX_R(T_, U_, N_) do { if (run_stop) LAUNCH_EX(T_, U_, N_, true); else LAUNCH_EX(T_, U_, N_, false); } while (0)
            if (exch_medium) LAUNCH_EX_R(EXM_THREADS, EXM_UNROLL, EXM_NPT);
            else if (exch_small) LAUNCH_EX_R(EXS_THREADS, EXS_UNROLL, EXS_NPT);
            else LAUNCH_EX_R(EX_THREADS, EX_UNROLL, EX_NPT);
#undef LAUNCH_EX_R
#undef LAUNCH_EX
            delta_var_valid = false;
        }
        prof_end(BDDMMA_K_FINISH_DELTA);
        HIPCHK(hipGetLastError());
        return BDDMMA_OK;
    }

    // ---------------------------------------------------------------------------- SolverBase
    int forward_run() override
    {
        if (fwd_valid) return BDDMMA_OK;
        HIPCHK(hipSetDevice(device));
        int rc = launch_fwd<FWD_PLAIN>(nullptr, REAL(0), BDDMMA_K_OTHER);
        if (rc) return rc;
        fwd_valid = true;
        return BDDMMA_OK;
    }
    int backward_run() override
    {
        if (bwd_valid) return BDDMMA_OK;
        HIPCHK(hipSetDevice(device));
        int rc = launch_bwd<BWD_PLAIN>(nullptr, REAL(0), BDDMMA_K_OTHER);
        if (rc) return rc;
        bwd_valid = true;
        return BDDMMA_OK;
    }
    int lower_bound(double* lb) override
    {
        int rc = backward_run();
        if (rc) return rc;
        if (lb_cached) {
            *lb = lb_cache;
            return BDDMMA_OK;
        }
        // no copy-engine round trip: one block writes the 8 bytes to pinned host memory, the host waits for the stream
        if ((rc = launch_bound(0))) return rc;
        if ((rc = wait_bound(0))) return rc;
        *lb = lb_cache = ((volatile double*)h_lb)[0];
        lb_cached = true;
        return BDDMMA_OK;
    }
    // The reduce kernel writes the bound and, behind it, a sequence number into pinned host memory; the host polls the number.  Waiting
    // for the stream instead goes through an interrupt: measured 26 us of idle GPU per bound read in the L-BFGS loop (tools/gaps.sh), which
    // reads 1.4 bounds per iteration.  After 2 ms without the number (a long queue ahead, or a fault) it falls back to the blocking wait.
    int launch_bound(int k)
    {
        lb_seq_expect[k] = ++lb_seq_next;
        hipLaunchKernelGGL(k_lb_reduce, dim3(1), dim3(1024), 0, stream, d_lb_partial, nb_.n_packs + wb_.n_packs + hb_.n_packs, d_lb + k, d_lb_seq + k, lb_seq_expect[k]);
        HIPCHK(hipGetLastError());
        return BDDMMA_OK;
    }
    int wait_bound(int k)
    {
        volatile uint64_t* w = h_lb_seq + k;
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t spins = 0;
        while (*w != lb_seq_expect[k]) {
            cpu_relax();
            if ((++spins & 1023u) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2e-3) {
                HIPCHK(hipStreamSynchronize(stream));
                if (*w != lb_seq_expect[k]) { err = "lower bound: the reduce kernel did not report"; return BDDMMA_ERR_DEVICE; }
                break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        return BDDMMA_OK;
    }
    int lower_bound_enqueue(int slot) override
    {
        if (slot < 0 || slot > 1) { err = "lower_bound_enqueue: slot must be 0 or 1"; return BDDMMA_ERR_INVALID_ARGUMENT; }
        int rc = backward_run();
        if (rc) return rc;
        lb_slot_gen[slot] = lb_gen;
        if (lb_cached) {
            lb_slot_known[slot] = true;
            lb_slot_value[slot] = lb_cache;
            return BDDMMA_OK;
        }
        if ((rc = launch_bound(1 + slot))) return rc;
        lb_slot_known[slot] = false;
        lb_slot_enqueued[slot] = true;
        return BDDMMA_OK;
    }
    int lower_bound_fetch(int slot, double* lb) override
    {
        if (slot < 0 || slot > 1) { err = "lower_bound_fetch: slot must be 0 or 1"; return BDDMMA_ERR_INVALID_ARGUMENT; }
        if (!lb_slot_known[slot]) {
            if (!lb_slot_enqueued[slot]) { err = "lower_bound_fetch without lower_bound_enqueue"; return BDDMMA_ERR_STATE; }
            int rc = wait_bound(1 + slot);
            if (rc) return rc;
            lb_slot_value[slot] = ((volatile double*)h_lb)[1 + slot];
            lb_slot_known[slot] = true;
        }
        *lb = lb_slot_value[slot];
        if (lb_slot_gen[slot] == lb_gen && bwd_valid) {  // no backward launch since: this is the bound of the current costs
            lb_cache = *lb;
            lb_cached = true;
        }
        return BDDMMA_OK;
    }
    int lower_bound_per_bdd(void* out, int on_device) override
    {
        int rc = backward_run();
        if (rc) return rc;
        REAL* tmp = nullptr;
        REAL* dst = (REAL*)out;
        if (!on_device) { HIPCHK(hipMalloc((void**)&tmp, n_bdds * sizeof(REAL))); dst = tmp; }
        hipLaunchKernelGGL((k_lb_per_bdd<REAL>), dim3(cdiv(n_bdds, 256)), dim3(256), 0, stream, d_T, d_root_slot, dst, (uint32_t)n_bdds);
        if (!on_device) {
            hipError_t e = hipMemcpyAsync(out, tmp, n_bdds * sizeof(REAL), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            (void)hipFree(tmp);
            HIPCHK(e);
        }
        return BDDMMA_OK;
    }

    int mma_forward(REAL omega, const REAL* delta_lay)
    {
        int rc;
        if (!bwd_valid && (rc = backward_run())) return rc;  // bdd_cuda_parallel_mma.cu:211-212
        rc = launch_fwd<FWD_SOLVE>(delta_lay, omega, BDDMMA_K_FORWARD_MM);
        x_layer_valid = false;  // the forward sweep rewrites the deferred values by entry only
        if (rc) return rc;
        fwd_valid = true;
        bwd_valid = false;
        return BDDMMA_OK;
    }
    int mma_backward(REAL omega, const REAL* delta_lay)
    {
        if (!fwd_valid) {
            err = "backward_mm requires a valid forward state (call forward_mm first)";  // assert at :304
            return BDDMMA_ERR_STATE;
        }
        int rc = launch_bwd<BWD_SOLVE>(delta_lay, omega, BDDMMA_K_BACKWARD_MM);
        if (rc) return rc;
        x_layer_valid = d_x_layer != nullptr;
        fwd_valid = false;
        bwd_valid = true;
        return BDDMMA_OK;
    }
    // ---- instances that fit one workgroup: n iterations inside one launch (kernels/small.hpp).  Not while event pairs are wanted per launch
    // (profiling) or an L-BFGS wrapper wants x in layer order from the backward sweeps.
    bool small_usable() const { return small_ok && !profiling && d_x_layer == nullptr; }
    int launch_small(REAL omega, uint32_t n, const RunStep& rstep)
    {
        int rc;
        if (!bwd_valid && (rc = backward_run())) return rc;  // bdd_cuda_parallel_mma.cu:211-212, as mma_forward
        lb_cached = false;
        ++lb_gen;
        const DevPtrs<REAL> d = ptrs(d_delta_lay);
        const PackDev pk = pdev(nb_, 0, 0);
#define LAUNCH_SMALL(NW_)                                                                                                                       \
    if (small_rl) hipLaunchKernelGGL((k_iterate_small<REAL, NW_, true>), dim3(1), dim3(64 * NW_), small_lds, stream, small, d, pk, omega, n, rstep); \
    else hipLaunchKernelGGL((k_iterate_small<REAL, NW_, false>), dim3(1), dim3(64 * NW_), small_lds, stream, small, d, pk, omega, n, rstep)
        switch (small_nw) { case 1: LAUNCH_SMALL(1); break; case 2: LAUNCH_SMALL(2); break; case 4: LAUNCH_SMALL(4); break; case 8: LAUNCH_SMALL(8); break; default: LAUNCH_SMALL(16); break; }
#undef LAUNCH_SMALL
        HIPCHK(hipGetLastError());
        // the state the last iteration's four launches leave (mma_forward, exchange, mma_backward, exchange)
        x_layer_valid = false;
        fwd_valid = false;
        bwd_valid = true;
        delta_var_valid = false;
        return BDDMMA_OK;
    }
    int iterations(double omega, uint64_t n) override
    {
        HIPCHK(hipSetDevice(device));
        if (!small_usable() || run_stop) {
            for (uint64_t i = 0; i < n; ++i)
                if (int rc = iteration(omega)) return rc;
            return BDDMMA_OK;
        }
        while (n) {
            const uint32_t chunk = (uint32_t)std::min<uint64_t>(n, 1u << 14);   // <= ~60 ms per launch: other processes' queues get the GPU in between
            if (int rc = launch_small((REAL)omega, chunk, RunStep{})) return rc;
            n -= chunk;
        }
        return BDDMMA_OK;
    }
    int iteration(double omega) override
    {
        HIPCHK(hipSetDevice(device));
        int rc;
        if (small_usable() && !run_stop) return launch_small((REAL)omega, 1, RunStep{});
        prof_active = profiling && (prof_iter++ % prof_stride == 0);
        if ((rc = mma_forward((REAL)omega, d_delta_lay))) return rc;
        if ((rc = exchange())) return rc;
        if ((rc = mma_backward((REAL)omega, d_delta_lay))) return rc;
        if ((rc = exchange(true))) return rc;
        return BDDMMA_OK;
    }
    // run_solver for the plain MMA iteration with the termination tests on the device (kernels.hpp: run_ctl_step).  The host
    // keeps up to `window` iterations queued and reads the published bounds; when the tests fire, the launches queued behind that
    // iteration return at once, so the state and the iteration count are the ones of the reference's sequential loop
    // (run_solver_util.h:40-73).  The wall-clock limit is tested on the host after every iteration it sees complete, as the reference
    // does; close to the limit the window shrinks to one iteration, so no iteration is queued that the sequential loop would not run.
    int run_plain(uint64_t max_iter, double tolerance, double slope, double time_limit, int verbose, bddmma_run_result* res) override
    {
        HIPCHK(hipSetDevice(device));
        const auto t0 = std::chrono::steady_clock::now();
        auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
        if (!d_run_ctl) {
            HIPCHK(hipMalloc((void**)&d_run_ctl, sizeof(RunCtl)));
            HIPCHK(hipHostMalloc((void**)&h_run, sizeof(RunHost), hipHostMallocMapped | hipHostMallocCoherent));
            HIPCHK(hipHostGetDevicePointer((void**)&d_run_host, h_run, 0));
        }
        double lb_initial;
        int rc = lower_bound(&lb_initial);
        if (rc) return rc;
        if (verbose) std::printf("[bdd solver] initial lower bound = %.10g, time = %.3f s\n", lb_initial, elapsed());
        RunCtl c{};
        c.stop = RUN_NOT_STOPPED;
        c.lb_initial = lb_initial;
        c.lb_first = std::numeric_limits<double>::max();
        c.lb_post = lb_initial;
        c.tolerance = tolerance;
        c.slope = slope;
        c.time_limit = time_limit;
        std::memset((void*)h_run, 0, sizeof(RunHost));
        run_step = RunStep{d_lb_partial, nb_.n_packs + wb_.n_packs + hb_.n_packs, d_run_ctl, d_run_host};
        HIPCHK(hipMemcpyAsync(d_run_ctl, &c, sizeof(RunCtl), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(k_run_begin, dim3(1), dim3(1), 0, stream, d_run_ctl, (uint64_t)(elapsed() * RUN_TICKS_PER_SECOND));
        HIPCHK(hipStreamSynchronize(stream));  // c is a stack object
        constexpr uint64_t WINDOW = 6;  // iterations queued ahead of the last bound seen (< RUN_RING)
        volatile RunHost* hr = h_run;
        uint64_t queued = 0, seen = 0;
        uint32_t idle_spins = 0;
        double idle_since = 0.0;
        int reason = 0;
        double lb_post = lb_initial;
        run_stop = &d_run_ctl->stop;
        while (true) {
            // (the wall-clock limit is tested on the device with the other criteria, run_ctl_step: iterations queued behind the one that
            // crossed it return at once like those behind any other stop)
            // instances that fit one workgroup: SMALL_CHUNK iterations per launch, the tests inside the kernel after each (k_iterate_small); at most
            // two chunks are outstanding, fewer bounds than the ring of published ones holds
            constexpr uint64_t SMALL_CHUNK = 64;
            static_assert(2 * SMALL_CHUNK < RUN_RING, "published bounds must not wrap before the host has read them");
            const bool fused = small_usable();
            const uint64_t window = fused ? SMALL_CHUNK + 1 : WINDOW;
            bool launched = false;
            while (queued < max_iter && queued - seen < window) {
                run_iter = (uint32_t)std::min<uint64_t>(queued, RUN_NOT_STOPPED - 1);
                const uint64_t n_launch = fused ? std::min<uint64_t>(SMALL_CHUNK, max_iter - queued) : 1;
                // the sequential path's last launch also reduces the bound and runs the tests (exchange(true))
                rc = fused ? launch_small(REAL(0.5), (uint32_t)n_launch, run_step) : iteration(0.5);
                if (rc) {  // launches of this and earlier iterations may be in flight: drain them before the gate goes away
                    (void)hipStreamSynchronize(stream);
                    run_stop = nullptr;
                    return rc;
                }
                queued += n_launch;
                launched = true;
            }
            if (queued == 0) break;  // max_iter == 0
            const uint64_t st = hr->state;
            const uint64_t done = st & ((1ull << 56) - 1);
            const int dev_reason = (int)(st >> 56);
            bool stop = false;
            const bool progressed = done > seen;
            for (; seen < done && !stop; ++seen) {
                lb_post = hr->lb[seen % RUN_RING];
                if (verbose) std::printf("[bdd solver] iteration %llu, lower bound = %.10g, time = %.3f s\n", (unsigned long long)seen, lb_post, elapsed());
            }
            if (stop) break;
            if (dev_reason && seen == done) { reason = dev_reason; break; }
            if (seen == max_iter) break;
            if (!launched && !progressed) {
                // nothing new (no launch in this pass, no bound published since the last one): spin for the first millisecond (an iteration is 20-250 us), then poll every 50 us without holding a core.
                // After 0.2 s without a published bound make sure the device is still alive: a blocking wait for the stream (harmless
                // if an iteration is simply that long) — still nothing then means the queued launches were lost, which the checks
                // behind the loop report.  (hipStreamQuery is not used for this: it was seen to report an idle stream with launches
                // still queued, which ended runs early.)  A pass that SAW progress is not idle, whatever the clock says: through round 5 the
                // test was "no launch in this pass", so a pass that found the window full at its start and all of it done at its read —
                // with a 0.2 s old idle stamp, which takes a GPU shared by a dozen processes — synchronised an empty stream, found "nothing
                // new" and ended the run early with reason 0 (soak of round 6: 1 of 32 loops; tests/tools/stress_run_solver.py: 15 of 3 200).
                const double now = elapsed();
                if (idle_spins++ == 0) idle_since = now;
                if (now - idle_since > 0.2) {
                    const hipError_t e = hipStreamSynchronize(stream);
                    if (e != hipSuccess) {
                        run_stop = nullptr;
                        err = std::string("run_solver: ") + hipGetErrorString(e);
                        return BDDMMA_ERR_DEVICE;
                    }
                    if ((hr->state & ((1ull << 56) - 1)) == seen && seen < queued) break;   // launches are outstanding and the idle stream has not run them
                    idle_spins = 0;
                } else if (now - idle_since > 1e-3) {
                    std::this_thread::sleep_for(std::chrono::microseconds(50));
                } else {
                    cpu_relax();
                }
            } else {
                idle_spins = 0;
            }
        }
        run_stop = nullptr;
        HIPCHK(hipStreamSynchronize(stream));
        // iterations that were still in flight when the time limit was seen have run too (at most `window` - 1; they are counted)
        {
            const uint64_t st = hr->state;
            const uint64_t done = st & ((1ull << 56) - 1);
            for (; seen < done; ++seen) lb_post = hr->lb[seen % RUN_RING];
        }
        if (reason == 0 && seen < queued) {
            err = "run_solver: queued iterations did not complete";
            return BDDMMA_ERR_DEVICE;
        }
        if (seen > 0) {  // the device's bound of the last iteration that ran is lower_bound() of the state it left (same sum, same order)
            lb_cache = lb_post;
            lb_cached = true;
        }
        if (verbose) std::printf("[bdd solver] final lower bound = %.10g\n", lb_post);
        if (res) {
            res->iterations = seen;
            res->lb_initial = lb_initial;
            res->lb_final = lb_post;
            res->seconds = elapsed();
            res->stop_reason = reason;
        }
        return BDDMMA_OK;
    }
    int explicit_mm(bool forward, double omega, void* delta, int on_device)
    {
        HIPCHK(hipSetDevice(device));
        const size_t bytes = 2 * n_vars * sizeof(REAL);
        HIPCHK(hipMemcpyAsync(d_delta_c, delta, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
        launch_bcast(d_delta_c, d_delta_lay_c);  // the caller's delta, as given, to every layer
        int rc = forward ? mma_forward((REAL)omega, d_delta_lay_c) : mma_backward((REAL)omega, d_delta_lay_c);
        if (rc) return rc;
        launch_reduce_raw(d_delta_c);
        HIPCHK(hipMemcpyAsync(delta, d_delta_c, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return BDDMMA_OK;
    }
    int forward_mm(double omega, void* delta, int on_device) override { return explicit_mm(true, omega, delta, on_device); }
    int backward_mm(double omega, void* delta, int on_device) override { return explicit_mm(false, omega, delta, on_device); }

    int normalize_delta(void* delta, int on_device) override
    {
        HIPCHK(hipSetDevice(device));
        const size_t bytes = 2 * n_vars * sizeof(REAL);
        REAL* p = (REAL*)delta;
        if (!on_device) { p = d_delta_c; HIPCHK(hipMemcpyAsync(p, delta, bytes, hipMemcpyHostToDevice, stream)); }
        hipLaunchKernelGGL((k_normalize_delta<REAL>), dim3(cdiv(2 * n_vars, 256)), dim3(256), 0, stream, p, d_nbdds, (uint32_t)(2 * n_vars));
        if (!on_device) HIPCHK(hipMemcpyAsync(delta, p, bytes, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return BDDMMA_OK;
    }
    int distribute_delta() override
    {
        HIPCHK(hipSetDevice(device));
        hipLaunchKernelGGL((k_distribute_delta<REAL>), dim3(cdiv(n_layers, 256)), dim3(256), 0, stream, d_lo, d_hi, d_mm_binned, d_lpos, (uint32_t)n_layers);
        x_layer_valid = false;
        HIPCHK(hipMemsetAsync(d_delta_var, 0, 2 * n_vars * sizeof(REAL), stream));  // bdd_cuda_base.cu:1428
        HIPCHK(hipMemsetAsync(d_delta_lay, 0, 2 * n_layers * sizeof(REAL), stream));
        delta_var_valid = true;
        costs_changed();
        HIPCHK(hipGetLastError());
        return BDDMMA_OK;
    }
    int get_delta(void* out, int on_device) override
    {
        HIPCHK(hipSetDevice(device));
        if (!delta_var_valid) {
            hipLaunchKernelGGL((k_delta_var_from_lay<REAL>), dim3(cdiv(n_vars, 256)), dim3(256), 0, stream, d_delta_lay, d_var_ptr, d_vpos,
                               d_delta_var, (uint32_t)n_vars);
            delta_var_valid = true;
        }
        HIPCHK(hipMemcpyAsync(out, d_delta_var, 2 * n_vars * sizeof(REAL), on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return BDDMMA_OK;
    }

    int set_delta(const void* in, int on_device) override
    {
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipMemcpyAsync(d_delta_var, in, 2 * n_vars * sizeof(REAL), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
        launch_bcast(d_delta_var, d_delta_lay);
        delta_var_valid = true;
        HIPCHK(hipStreamSynchronize(stream));
        return BDDMMA_OK;
    }

    template <typename TIN>
    int update_both(const void* lo, uint64_t n_lo, const void* hi, uint64_t n_hi, int on_device)
    {
        if (n_lo == 0 && n_hi == 0) return BDDMMA_OK;
        if (n_lo > n_vars || n_hi > n_vars) { err = "cost vector longer than nr_variables()"; return BDDMMA_ERR_INVALID_ARGUMENT; }
        const TIN *dlo = (const TIN*)lo, *dhi = (const TIN*)hi;
        TIN* tmp = nullptr;
        if (!on_device) {  // one staging buffer for both sides
            HIPCHK(hipMalloc((void**)&tmp, (n_lo + n_hi) * sizeof(TIN)));
            hipError_t e = hipSuccess;
            if (n_lo) e = hipMemcpyAsync(tmp, lo, n_lo * sizeof(TIN), hipMemcpyHostToDevice, stream);
            if (n_hi && e == hipSuccess) e = hipMemcpyAsync(tmp + n_lo, hi, n_hi * sizeof(TIN), hipMemcpyHostToDevice, stream);
            if (e != hipSuccess) { (void)hipFree(tmp); HIPCHK(e); }
            dlo = tmp;
            dhi = tmp + n_lo;
        }
        if (!d_cost_q) {
            int rc;
            if ((rc = dalloc(&d_cost_q, n_vars)) || (rc = dalloc(&d_cost_flags, n_vars))) { if (tmp) (void)hipFree(tmp); return rc; }
        }
        hipLaunchKernelGGL((k_cost_quotients<TIN>), dim3(cdiv(n_vars, 256)), dim3(256), 0, stream, d_cost_q, d_cost_flags, d_nbdds, dlo, n_lo, dhi, n_hi,
                           (uint32_t)n_vars);
        hipLaunchKernelGGL((k_update_costs<REAL>), dim3(cdiv(n_layers, 256)), dim3(256), 0, stream, d_lohi, d_var, (const CostQuot*)d_cost_q,
                           (const uint8_t*)d_cost_flags, (uint32_t)(n_lo != 0), (uint32_t)(n_hi != 0), (uint32_t)n_layers);
        if (tmp) { HIPCHK(hipStreamSynchronize(stream)); (void)hipFree(tmp); }
        return BDDMMA_OK;
    }
    int update_costs(const void* lo, uint64_t n_lo, const void* hi, uint64_t n_hi, int elem_precision, int on_device) override
    {
        HIPCHK(hipSetDevice(device));
        int rc = elem_precision == BDDMMA_F64 ? update_both<double>(lo, n_lo, hi, n_hi, on_device) : update_both<float>(lo, n_lo, hi, n_hi, on_device);
        if (rc) return rc;
        costs_changed();
        HIPCHK(hipGetLastError());
        return BDDMMA_OK;
    }
    int set_cost(double c, uint64_t var) override
    {
        if (var >= n_vars) { err = "variable out of range"; return BDDMMA_ERR_INVALID_ARGUMENT; }
        HIPCHK(hipSetDevice(device));
        const uint32_t k0 = h_var_ptr[var], k1 = h_var_ptr[var + 1];
        if (k1 > k0) {
            const REAL cc = REAL(c / (double)(k1 - k0));
            hipLaunchKernelGGL((k_set_cost<REAL>), dim3(cdiv(k1 - k0, 64)), dim3(64), 0, stream, d_hi, d_var_layers, k0, k1, cc);
        }
        costs_changed();
        HIPCHK(hipGetLastError());
        return BDDMMA_OK;
    }
    int copy_out(void* dst, const void* src, size_t bytes, int on_device) const
    {
        std::string& err = const_cast<std::string&>(this->err);
        if (!dst) return BDDMMA_OK;
        HIPCHK(hipMemcpyAsync(dst, src, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return BDDMMA_OK;
    }
    int get_solver_costs(void* lo, void* hi, void* mm, int on_device) override
    {
        int rc;
        std::string& err = const_cast<std::string&>(this->err);
        HIPCHK(hipSetDevice(device));
        const dim3 g(cdiv(n_layers, 256)), b(256);
        if (lo) {
            hipLaunchKernelGGL((k_strided_copy<REAL>), g, b, 0, stream, d_tmp0, 1u, (const REAL*)d_lo, 2u, (uint32_t)n_layers);
            if ((rc = copy_out(lo, d_tmp0, n_layers * sizeof(REAL), on_device))) return rc;
        }
        if (hi) {
            hipLaunchKernelGGL((k_strided_copy<REAL>), g, b, 0, stream, d_tmp0, 1u, (const REAL*)d_hi, 2u, (uint32_t)n_layers);
            if ((rc = copy_out(hi, d_tmp0, n_layers * sizeof(REAL), on_device))) return rc;
        }
        if (!mm) return BDDMMA_OK;
        hipLaunchKernelGGL((k_entries_to_layers<REAL>), dim3(cdiv(n_layers, 256)), dim3(256), 0, stream, d_mm_binned, d_lpos, d_tmp0, (uint32_t)n_layers);
        return copy_out(mm, d_tmp0, n_layers * sizeof(REAL), on_device);
    }
    int set_solver_costs(const void* lo, const void* hi, const void* mm, int on_device) override
    {
        const hipMemcpyKind k = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        HIPCHK(hipSetDevice(device));
        const dim3 g(cdiv(n_layers, 256)), b(256);
        if (lo) {
            HIPCHK(hipMemcpyAsync(d_tmp0, lo, n_layers * sizeof(REAL), k, stream));
            hipLaunchKernelGGL((k_strided_copy<REAL>), g, b, 0, stream, d_lo, 2u, (const REAL*)d_tmp0, 1u, (uint32_t)n_layers);
        }
        if (hi) {
            HIPCHK(hipMemcpyAsync(d_tmp0, hi, n_layers * sizeof(REAL), k, stream));
            hipLaunchKernelGGL((k_strided_copy<REAL>), g, b, 0, stream, d_hi, 2u, (const REAL*)d_tmp0, 1u, (uint32_t)n_layers);
        }
        if (mm) {
            x_layer_valid = false;
            HIPCHK(hipMemcpyAsync(d_tmp0, mm, n_layers * sizeof(REAL), k, stream));
            hipLaunchKernelGGL((k_layers_to_entries<REAL>), dim3(cdiv(n_layers, 256)), dim3(256), 0, stream, d_tmp0, d_lpos, d_mm_binned, (uint32_t)n_layers);
        }
        HIPCHK(hipStreamSynchronize(stream));
        costs_changed();
        return BDDMMA_OK;
    }
    int primal_objective_vec(void* out, int on_device) override
    {
        HIPCHK(hipSetDevice(device));
        REAL* dst = on_device ? (REAL*)out : d_delta_c;  // 2V scratch is large enough
        hipLaunchKernelGGL((k_primal_objective<REAL>), dim3(cdiv(n_vars, 256)), dim3(256), 0, stream, d_lo, d_hi, d_var_ptr, d_var_layers, dst, (uint32_t)n_vars);
        if (!on_device) return copy_out(out, dst, n_vars * sizeof(REAL), 0);
        HIPCHK(hipStreamSynchronize(stream));
        return BDDMMA_OK;
    }
    int min_marginals(int sorted, int32_t* var, void* mm0, void* mm1, int on_device) override
    {
        HIPCHK(hipSetDevice(device));
        int rc;
        if ((rc = forward_run())) return rc;  // bdd_cuda_base.cu:720
        if ((rc = launch_bwd<BWD_MARGINALS>(nullptr, REAL(0), BDDMMA_K_OTHER))) return rc;
        bwd_valid = true;
        const hipMemcpyKind k = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        if (!sorted) {
            if (var) HIPCHK(hipMemcpyAsync(var, d_var, n_layers * sizeof(int32_t), k, stream));
            if (mm0) HIPCHK(hipMemcpyAsync(mm0, d_tmp0, n_layers * sizeof(REAL), k, stream));
            if (mm1) HIPCHK(hipMemcpyAsync(mm1, d_tmp1, n_layers * sizeof(REAL), k, stream));
            HIPCHK(hipStreamSynchronize(stream));
            return BDDMMA_OK;
        }
        // gather by primal_variable_sorting_order_ (:737-746)
        REAL *s0 = nullptr, *s1 = nullptr;
        int32_t* sv = nullptr;
        HIPCHK(hipMalloc((void**)&s0, std::max<uint64_t>(n_layers, 1) * sizeof(REAL)));
        HIPCHK(hipMalloc((void**)&s1, std::max<uint64_t>(n_layers, 1) * sizeof(REAL)));
        HIPCHK(hipMalloc((void**)&sv, std::max<uint64_t>(n_layers, 1) * sizeof(int32_t)));
        const dim3 g(cdiv(n_layers, 256)), b(256);
        hipLaunchKernelGGL((k_gather<REAL>), g, b, 0, stream, d_tmp0, d_var_layers, s0, (uint32_t)n_layers);
        hipLaunchKernelGGL((k_gather<REAL>), g, b, 0, stream, d_tmp1, d_var_layers, s1, (uint32_t)n_layers);
        hipLaunchKernelGGL(k_gather_var, g, b, 0, stream, d_var, d_var_layers, sv, (uint32_t)n_layers);
        hipError_t e = hipSuccess;
        if (var && e == hipSuccess) e = hipMemcpyAsync(var, sv, n_layers * sizeof(int32_t), k, stream);
        if (mm0 && e == hipSuccess) e = hipMemcpyAsync(mm0, s0, n_layers * sizeof(REAL), k, stream);
        if (mm1 && e == hipSuccess) e = hipMemcpyAsync(mm1, s1, n_layers * sizeof(REAL), k, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(s0); (void)hipFree(s1); (void)hipFree(sv);
        HIPCHK(e);
        return BDDMMA_OK;
    }
    int min_marginal_diff(void* out, int on_device) override
    {
        HIPCHK(hipSetDevice(device));
        int rc;
        if ((rc = forward_run())) return rc;
        if ((rc = launch_bwd<BWD_MARGINALS>(nullptr, REAL(0), BDDMMA_K_OTHER))) return rc;
        bwd_valid = true;
        REAL* dst = on_device ? (REAL*)out : d_tmp0;  // in place over mm0 when the result goes to the host
        hipLaunchKernelGGL((k_diff<REAL, REAL>), dim3(cdiv(n_layers, 256)), dim3(256), 0, stream, dst, (const REAL*)d_tmp1, (const REAL*)d_tmp0, (uint32_t)n_layers);
        if (!on_device) return copy_out(out, dst, n_layers * sizeof(REAL), 0);
        HIPCHK(hipStreamSynchronize(stream));
        return BDDMMA_OK;
    }
    int bdds_solution(int sorted, char* sol, int on_device) override
    {
        HIPCHK(hipSetDevice(device));
        int rc;
        if ((rc = backward_run())) return rc;
        HIPCHK(hipMemsetAsync(d_sol, 0, n_layers, stream));
        if ((rc = launch_fwd<FWD_SOLUTION>(nullptr, REAL(0), BDDMMA_K_OTHER))) return rc;
        // (fwd_valid stays what it was: the solution sweep recomputes the costs-from-root on the fly and does not store them)
        const hipMemcpyKind k = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        if (!sorted) {
            HIPCHK(hipMemcpyAsync(sol, d_sol, n_layers, k, stream));
            HIPCHK(hipStreamSynchronize(stream));
            return BDDMMA_OK;
        }
        char* s = nullptr;
        HIPCHK(hipMalloc((void**)&s, std::max<uint64_t>(n_layers, 1)));
        hipLaunchKernelGGL((k_gather<char>), dim3(cdiv(n_layers, 256)), dim3(256), 0, stream, d_sol, d_var_layers, s, (uint32_t)n_layers);
        hipError_t e = hipMemcpyAsync(sol, s, n_layers, k, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(s);
        HIPCHK(e);
        return BDDMMA_OK;
    }
    int bdds_solution_async(char* dev_out, int prezeroed) override
    {
        HIPCHK(hipSetDevice(device));
        int rc;
        if ((rc = backward_run())) return rc;
        if (!prezeroed) HIPCHK(hipMemsetAsync(dev_out, 0, n_layers, stream));
        char* const own = d_sol;
        d_sol = dev_out;  // ptrs() hands d_sol to the sweep
        rc = launch_fwd<FWD_SOLUTION>(nullptr, REAL(0), BDDMMA_K_OTHER);
        d_sol = own;
        if (rc) return rc;
        return BDDMMA_OK;
    }
    int lbfgs_views(LbfgsViews* out) override
    {
        HIPCHK(hipSetDevice(device));
        int rc;
        if (!d_x_layer && (rc = dalloc(&d_x_layer, n_layers))) return rc;
        if (!x_layer_valid) {  // the first call, or something other than a backward solve sweep has changed costs / deferred values since
            hipLaunchKernelGGL((k_net_costs<REAL>), dim3(cdiv(n_layers, 256)), dim3(256), 0, stream, d_lo, d_hi, d_mm_binned, d_lpos, d_x_layer, (uint32_t)n_layers);
            HIPCHK(hipGetLastError());
            x_layer_valid = true;
        }
        *out = LbfgsViews{d_x_layer};
        return BDDMMA_OK;
    }
    int net_solver_costs(void* out, int on_device) override
    {
        HIPCHK(hipSetDevice(device));
        REAL* dst = on_device ? (REAL*)out : d_tmp0;
        hipLaunchKernelGGL((k_net_costs<REAL>), dim3(cdiv(n_layers, 256)), dim3(256), 0, stream, d_lo, d_hi, d_mm_binned, d_lpos, dst, (uint32_t)n_layers);
        if (!on_device) return copy_out(out, dst, n_layers * sizeof(REAL), 0);
        HIPCHK(hipGetLastError());
        return BDDMMA_OK;
    }
    int make_dual_feasible(void* g, int on_device) override
    {
        HIPCHK(hipSetDevice(device));
        REAL* p = (REAL*)g;
        if (!on_device) { p = d_tmp0; HIPCHK(hipMemcpyAsync(p, g, n_layers * sizeof(REAL), hipMemcpyHostToDevice, stream)); }
        hipLaunchKernelGGL((k_make_dual_feasible<REAL>), dim3(cdiv(n_vars, 256)), dim3(256), 0, stream, p, d_var_ptr, d_var_layers, (uint32_t)n_vars);
        if (!on_device) return copy_out(g, p, n_layers * sizeof(REAL), 0);
        HIPCHK(hipGetLastError());
        return BDDMMA_OK;
    }
    int gradient_step(const void* g, double step, int on_device) override
    {
        HIPCHK(hipSetDevice(device));
        const REAL* p = (const REAL*)g;
        if (!on_device) { HIPCHK(hipMemcpyAsync(d_tmp0, g, n_layers * sizeof(REAL), hipMemcpyHostToDevice, stream)); p = d_tmp0; }
        hipLaunchKernelGGL((k_gradient_step<REAL>), dim3(cdiv(n_layers, 256)), dim3(256), 0, stream, d_hi, p, REAL(step), (uint32_t)n_layers);
        costs_changed();
        HIPCHK(hipGetLastError());
        return BDDMMA_OK;
    }
    // make_dual_feasible + gradient_step for a device vector that is only applied, never read back (lbfgs.hip).
    // Large instances (>= 500 k layers; variant_flags bit 9 forces it, bit 10 forbids it): the vector goes through the staging tables
    // to entry order, is projected per bin in LDS and comes back to layer order together with the first step (kernels.hpp:
    // k_stage_transpose / k_project_entries) — then every further step is a plain stream.  Small ones: the per-variable means by gather ...
    REAL* d_proj_q = nullptr;
    REAL* d_proj_dir = nullptr;     // the projected vector in layer order (staged path)
    REAL* d_proj_ent = nullptr;     // ... and in entry order
    bool proj_staged = false;       // the last projection_means took the staged path
    bool proj_pending = false;      // ... and its way back to layer order is still to be done (with the first step)
    bool proj_attr_set = false;
    const void* proj_of = nullptr;  // the vector the last projection_means was called on
    bool use_staged_projection() const
    {
        if (opts_variant & 0x400u) return false;
        if (n_narrow_layers == 0) return false;
        return (opts_variant & 0x200u) != 0 || n_layers >= 500000;
    }
    template <int TO_LAYERS>
    void launch_stage_transpose(const REAL* in, REAL* out, REAL* lohi, REAL step)
    {
        const PackDev pk = pdev(nb_, 0);
        const uint32_t n_quads = (uint32_t)cdiv(nb_.n_packs, wpb);
        const uint32_t lds = wpb * stage_cap * (uint32_t)sizeof(REAL);
#define LAUNCH_T(W_) hipLaunchKernelGGL((k_stage_transpose<REAL, W_, TO_LAYERS>), dim3(n_quads), dim3(64 * W_), lds, stream, in, out, pk, (const uint32_t*)d_cs_entry, \
                                        (const uint16_t*)d_cs_slot, n_narrow_layers, (uint32_t)n_layers, lohi, step)
        switch (wpb) { case 1: LAUNCH_T(1); break; case 2: LAUNCH_T(2); break; case 4: LAUNCH_T(4); break; default: LAUNCH_T(8); break; }
#undef LAUNCH_T
    }
    int projection_means(const void* g) override
    {
        HIPCHK(hipSetDevice(device));
        int rc;
        proj_staged = use_staged_projection();
        proj_pending = false;
        proj_of = g;
        if (!proj_staged) {
            if (!d_proj_q && (rc = dalloc(&d_proj_q, n_vars))) return rc;
            hipLaunchKernelGGL((k_projection_means<REAL>), dim3(cdiv(n_vars, 256)), dim3(256), 0, stream, (const REAL*)g, d_var_ptr, d_var_layers, d_proj_q, (uint32_t)n_vars);
            HIPCHK(hipGetLastError());
            return BDDMMA_OK;
        }
        if (!d_proj_dir && (rc = dalloc(&d_proj_dir, n_layers))) return rc;
        if (!d_proj_ent && (rc = dalloc(&d_proj_ent, n_layers))) return rc;
        const uint32_t plds = vars_per_bin * (uint32_t)sizeof(double);
        if (!proj_attr_set) {
            if (plds > 64 * 1024) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_project_entries<REAL, EX_THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));
            proj_attr_set = true;
        }
        // layers -> entries 
        launch_stage_transpose<0>((const REAL*)g, d_proj_ent, (REAL*)nullptr, REAL(0));
        if (n_layers > n_narrow_layers)
            hipLaunchKernelGGL((k_layers_to_entries<REAL>), dim3(cdiv(n_layers - n_narrow_layers, 256)), dim3(256), 0, stream, (const REAL*)g + n_narrow_layers,
                               d_lpos + n_narrow_layers, d_proj_ent, (uint32_t)(n_layers - n_narrow_layers));
        // x_e -= mean over the entries of its variable
#define LAUNCH_P(T_) hipLaunchKernelGGL((k_project_entries<REAL, T_>), dim3(n_bins), dim3(T_), plds, stream, d_proj_ent, d_bin_ptr, d_bvar, d_nbdds, vars_per_bin, (uint32_t)n_vars, (uint32_t)n_layers)
        if (exch_small) LAUNCH_P(EXS_THREADS);
        else if (exch_medium) LAUNCH_P(EXM_THREADS);
        else LAUNCH_P(EX_THREADS);
#undef LAUNCH_P
        proj_pending = true;
        HIPCHK(hipGetLastError());
        return BDDMMA_OK;
    }
    // The same for a vector that is a linear combination of stored ones (the L-BFGS direction): formed inside the layers -> entries pass
    // (kernels.hpp: k_stage_lincomb).  Instances of narrow packs only; variant_flags bit 15: not offered (the wrapper writes the direction
    // and calls projection_means).
    bool projection_fuses_lincomb() const override { return use_staged_projection() && n_layers == n_narrow_layers && !(opts_variant & 0x8000u); }
    int projection_means_lincomb(const LinComb& lc, const void* tag) override
    {
        HIPCHK(hipSetDevice(device));
        int rc;
        if (!projection_fuses_lincomb() || lc.ns < 1 || lc.ns > LINCOMB_MAX) { err = "projection_means_lincomb: not available for this instance"; return BDDMMA_ERR_INVALID_ARGUMENT; }
        proj_staged = true;
        proj_pending = false;
        proj_of = tag;
        if (!d_proj_dir && (rc = dalloc(&d_proj_dir, n_layers))) return rc;
        if (!d_proj_ent && (rc = dalloc(&d_proj_ent, n_layers))) return rc;
        const uint32_t plds = vars_per_bin * (uint32_t)sizeof(double);
        if (!proj_attr_set) {
            if (plds > 64 * 1024) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_project_entries<REAL, EX_THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));
            proj_attr_set = true;
        }
        {
            const PackDev pk = pdev(nb_, 0);
            const uint32_t n_quads = (uint32_t)cdiv(nb_.n_packs, wpb);
            const uint32_t lds = wpb * stage_cap * (uint32_t)sizeof(REAL);
#define LAUNCH_LC(W_, NS_) hipLaunchKernelGGL((k_stage_lincomb<REAL, W_, NS_>), dim3(n_quads), dim3(64 * W_), lds, stream, lc, d_proj_ent, pk, (const uint32_t*)d_cs_entry, \
                                              (const uint16_t*)d_cs_slot, n_narrow_layers, (uint32_t)n_layers)
#define LAUNCH_LCW(W_) do { if (lc.ns == 5) LAUNCH_LC(W_, 5); else LAUNCH_LC(W_, 0); } while (0)
            switch (wpb) { case 1: LAUNCH_LCW(1); break; case 2: LAUNCH_LCW(2); break; case 4: LAUNCH_LCW(4); break; default: LAUNCH_LCW(8); break; }
#undef LAUNCH_LCW
#undef LAUNCH_LC
        }
#define LAUNCH_P(T_) hipLaunchKernelGGL((k_project_entries<REAL, T_>), dim3(n_bins), dim3(T_), plds, stream, d_proj_ent, d_bin_ptr, d_bvar, d_nbdds, vars_per_bin, (uint32_t)n_vars, (uint32_t)n_layers)
        if (exch_small) LAUNCH_P(EXS_THREADS);
        else if (exch_medium) LAUNCH_P(EXM_THREADS);
        else LAUNCH_P(EX_THREADS);
#undef LAUNCH_P
        proj_pending = true;
        HIPCHK(hipGetLastError());
        return BDDMMA_OK;
    }
    // ... then any number of steps along the projected vector
    int gradient_step_projected(const void* g, double step) override
    {
        HIPCHK(hipSetDevice(device));
        if (g != proj_of) { err = "gradient_step_projected: not the vector projection_means was called on"; return BDDMMA_ERR_INVALID_ARGUMENT; }
        if (proj_staged) {
            if (proj_pending) {
                // entries -> layers, with this step applied on the way
                launch_stage_transpose<1>((const REAL*)d_proj_ent, d_proj_dir, d_lohi, REAL(step));
                if (n_layers > n_narrow_layers) {
                    const uint32_t nw = (uint32_t)(n_layers - n_narrow_layers);
                    hipLaunchKernelGGL((k_entries_to_layers<REAL>), dim3(cdiv(nw, 256)), dim3(256), 0, stream, (const REAL*)d_proj_ent, d_lpos + n_narrow_layers,
                                       d_proj_dir + n_narrow_layers, nw);
                    hipLaunchKernelGGL((k_gradient_step<REAL>), dim3(cdiv(nw, 256)), dim3(256), 0, stream, d_hi + 2 * (size_t)n_narrow_layers,
                                       (const REAL*)d_proj_dir + n_narrow_layers, REAL(step), nw);
                }
                proj_pending = false;
            } else {
                hipLaunchKernelGGL((k_gradient_step<REAL>), dim3(cdiv(n_layers, 256)), dim3(256), 0, stream, d_hi, (const REAL*)d_proj_dir, REAL(step), (uint32_t)n_layers);
            }
            costs_changed();
            HIPCHK(hipGetLastError());
            return BDDMMA_OK;
        }
        if (!d_proj_q) { err = "gradient_step_projected without projection_means"; return BDDMMA_ERR_INVALID_ARGUMENT; }
        hipLaunchKernelGGL((k_gradient_step_projected<REAL>), dim3(cdiv(n_layers, 256)), dim3(256), 0, stream, d_hi, (const REAL*)g, d_proj_q, (const uint32_t*)d_var, REAL(step), (uint32_t)n_layers);
        costs_changed();
        HIPCHK(hipGetLastError());
        return BDDMMA_OK;
    }
    void* stream_handle() override { return (void*)stream; }
    int rounding_scratch(void** c0_dev, void** c1_dev) override
    {
        *c0_dev = d_delta_c;
        *c1_dev = d_delta_c + n_vars;
        return BDDMMA_OK;
    }
    int rounding_round(double delta, uint32_t round, uint32_t seed, uint32_t counts[4], char* sol_host, void* c0_host, void* c1_host,
                       bool apply_update, int* applied) override
    {
        HIPCHK(hipSetDevice(device));
        int rc;
        *applied = 0;
        if ((rc = distribute_delta())) return rc;                 // :265
        if ((rc = forward_run())) return rc;                      // min_marginals_cuda(), :266
        if ((rc = launch_bwd<BWD_MARGINALS>(nullptr, REAL(0), BDDMMA_K_OTHER))) return rc;
        bwd_valid = true;
        HIPCHK(hipMemsetAsync(d_counts, 0, 4 * sizeof(uint32_t), stream));
        REAL* c0 = d_delta_c;               // 2V scratch: [0,V) cost_delta_0, [V,2V) cost_delta_1
        REAL* c1 = d_delta_c + n_vars;
        hipLaunchKernelGGL((k_round_perturb<REAL>), dim3(cdiv(n_vars, 256)), dim3(256), 0, stream, d_tmp0, d_tmp1, d_var_ptr, d_var_layers,
                           c0, c1, d_sol, d_counts, (uint32_t)n_vars, delta, round, seed);
        HIPCHK(hipMemcpyAsync(counts, d_counts, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        if (c0_host) HIPCHK(hipMemcpyAsync(c0_host, c0, n_vars * sizeof(REAL), hipMemcpyDeviceToHost, stream));
        if (c1_host) HIPCHK(hipMemcpyAsync(c1_host, c1, n_vars * sizeof(REAL), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if ((uint64_t)counts[0] + counts[1] == n_vars) {          // all min-marginals agree: read off the solution, :295-305
            HIPCHK(hipMemcpyAsync(sol_host, d_sol, n_vars, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            return BDDMMA_OK;
        }
        *applied = 1;
        if (!apply_update) return BDDMMA_OK;                        // the caller applies it (through the L-BFGS wrapper)
        return update_costs(c0, n_vars, c1, n_vars, precision, 1);  // :327
    }
    // STREAM triad / copy over temporary BDDMMA_TRIAD_BYTES arrays (n4 is a multiple of 4 * grid * 256)
    int time_stream(bool copy, uint64_t reps, double* ms)
    {
        const uint64_t n4 = BDDMMA_TRIAD_BYTES / 16;
        stream_v4 *a = nullptr, *b = nullptr, *c = nullptr;
        HIPCHK(hipMalloc(&a, BDDMMA_TRIAD_BYTES));
        HIPCHK(hipMalloc(&b, BDDMMA_TRIAD_BYTES));
        HIPCHK(hipMalloc(&c, BDDMMA_TRIAD_BYTES));
        HIPCHK(hipMemsetAsync(b, 0, BDDMMA_TRIAD_BYTES, stream));
        HIPCHK(hipMemsetAsync(c, 0, BDDMMA_TRIAD_BYTES, stream));
        const unsigned grid = (unsigned)((n4 + 255) / 256);  // one vector per thread (kernels.hpp: k_stream)
        auto once = [&]() {
            if (copy) k_stream<true><<<grid, 256, 0, stream>>>(a, b, c, 3.0f, n4);
            else k_stream<false><<<grid, 256, 0, stream>>>(a, b, c, 3.0f, n4);
        };
        once();
        HIPCHK(hipEventRecord(ev_t0, stream));
        for (uint64_t i = 0; i < reps; ++i) once();
        HIPCHK(hipEventRecord(ev_t1, stream));
        HIPCHK(hipEventSynchronize(ev_t1));
        float f = 0.f;
        HIPCHK(hipEventElapsedTime(&f, ev_t0, ev_t1));
        *ms = f;
        HIPCHK(hipFree(a));
        HIPCHK(hipFree(b));
        HIPCHK(hipFree(c));
        return BDDMMA_OK;
    }

    int time_kernel(int kind, uint64_t reps, double* ms) override
    {
        HIPCHK(hipSetDevice(device));
        int rc = BDDMMA_OK;
        auto once = [&]() -> int {
            switch (kind) {
                case 0: return launch_fwd<FWD_PLAIN>(nullptr, REAL(0), BDDMMA_K_OTHER);
                case 1: return launch_bwd<BWD_PLAIN>(nullptr, REAL(0), BDDMMA_K_OTHER);
                case 2: return launch_fwd<FWD_SOLVE>(d_delta_lay, REAL(0.5), BDDMMA_K_OTHER);
                case 3: return launch_bwd<BWD_SOLVE>(d_delta_lay, REAL(0.5), BDDMMA_K_OTHER);
                case 4: return exchange();
                case 5: launch_bcast(d_delta_c, d_delta_lay_c); return BDDMMA_OK;
                default: err = "unknown kernel kind"; return BDDMMA_ERR_INVALID_ARGUMENT;
            }
        };
        if (kind == 6 || kind == 7) return time_stream(kind == 7, reps, ms);
        if ((rc = once())) return rc;  // warm-up
        HIPCHK(hipEventRecord(ev_t0, stream));
        for (uint64_t i = 0; i < reps; ++i)
            if ((rc = once())) return rc;
        HIPCHK(hipEventRecord(ev_t1, stream));
        HIPCHK(hipEventSynchronize(ev_t1));
        float f = 0.f;
        HIPCHK(hipEventElapsedTime(&f, ev_t0, ev_t1));
        *ms = f;
        costs_changed();
        x_layer_valid = false;
#ifdef BDDMMA_STAMPS
        if (const char* path = std::getenv("BDDMMA_STAMPS_FILE")) {  // one more launch with per-wave phase stamps (kernels.hpp: BDDMMA_STAMP)
            const size_t slots = 0x100000u + 65536u * 16u;
            unsigned long long* d_st = nullptr;
            HIPCHK(hipMalloc((void**)&d_st, slots * 8 * sizeof(unsigned long long)));
            HIPCHK(hipMemset(d_st, 0, slots * 8 * sizeof(unsigned long long)));
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_bddmma_stamps), &d_st, sizeof(d_st)));
            HIPCHK(hipStreamSynchronize(stream));
            rc = once();
            HIPCHK(hipStreamSynchronize(stream));
            unsigned long long* null_ = nullptr;
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_bddmma_stamps), &null_, sizeof(null_)));
            std::vector<unsigned long long> h(slots * 8);
            HIPCHK(hipMemcpy(h.data(), d_st, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            (void)hipFree(d_st);
            if (FILE* f2 = std::fopen((std::string(path) + "." + std::to_string(kind)).c_str(), "w")) {
                for (size_t sl = 0; sl < slots; ++sl)
                    if (h[sl * 8] != 0)
                        std::fprintf(f2, "%zu %llu %llu %llu %llu %llu\n", sl, h[sl * 8], h[sl * 8 + 1], h[sl * 8 + 2], h[sl * 8 + 3], h[sl * 8 + 4]);
                std::fclose(f2);
            }
            if (rc) return rc;
        }
#endif
        return BDDMMA_OK;
    }
};

}  // namespace bddmma

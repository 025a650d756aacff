/*
 * mma_oracle_impl.h — body of the CPU oracle, included once per precision by mma_oracle.c
 * with REAL and SUFFIX defined.
 *
 * TEST INFRASTRUCTURE ONLY (see mma_oracle.c header).
 *
 * Restates, BDD by BDD, the reference CPU solver
 *   LPMP::bdd_parallel_mma_base<bdd_branch_instruction<REAL,uint16_t>>
 * (reference: src/bdd_solver/bdd_parallel_mma_base.cpp, include/bdd_solver/bdd_branch_instruction.h).
 * Each function cites the reference lines it follows.  Node storage mirrors
 * bdd_branch_instruction_base {m, low_cost, high_cost, offset_low, offset_high}
 * (bdd_branch_instruction.h:13-24) except that offsets are uint32 instead of uint16.
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)
#define NODE CAT(onode, SUFFIX)
#define ORACLE CAT(oracle, SUFFIX)

typedef struct NODE {
    REAL m;
    REAL low_cost;
    REAL high_cost;
    uint32_t offset_low;
    uint32_t offset_high;
} NODE;

typedef struct ORACLE {
    size_t n_bdds, n_vars, n_nodes, n_layers;
    NODE* nodes;
    size_t* bdd_layer_ptr;   /* n_bdds+1: first layer of each BDD (bdd_variables_ rows) */
    size_t* layer_node_ptr;  /* n_layers+1: first node of each layer (bdd_variable::offset) */
    size_t* layer_var;       /* n_layers (bdd_variable::variable) */
    size_t* nr_bdds_per_var; /* n_vars */
    REAL* delta_in;          /* 2*n_vars, {lo,hi} interleaved; empty until first iteration() */
    REAL* delta_out;
    REAL* mm_last;           /* n_layers: what the layer itself gave up in its last mm pass (the GPU solver's deffered_mm_diff_) */
    int have_delta_in, have_delta_out;
    int mp_state;            /* message_passing_state: 0 none, 1 after_forward, 2 after_backward */
    int lb_valid;
    double lower_bound;
    int n_threads;
} ORACLE;

#define MP_NONE 0
#define MP_FWD 1
#define MP_BWD 2

/* bdd_branch_instruction.h:101-130 */
static inline void FN(backward_step)(NODE* n)
{
    if (n->offset_low == TERM0 || n->offset_low == TERM1)
        n->m = n->low_cost;
    else
        n->m = (n + n->offset_low)->m + n->low_cost;
    if (n->offset_high == TERM0 || n->offset_high == TERM1)
        n->m = RMIN(n->m, n->high_cost);
    else
        n->m = RMIN(n->m, (n + n->offset_high)->m + n->high_cost);
}

/* bdd_branch_instruction.h:132-147 */
static inline void FN(prepare_forward_step)(NODE* n)
{
    if (n->offset_low != TERM0 && n->offset_low != TERM1) (n + n->offset_low)->m = RINF;
    if (n->offset_high != TERM0 && n->offset_high != TERM1) (n + n->offset_high)->m = RINF;
}

/* bdd_branch_instruction.h:149-169 */
static inline void FN(forward_step)(NODE* n)
{
    if (n->offset_low != TERM0 && n->offset_low != TERM1) {
        NODE* c = n + n->offset_low;
        c->m = RMIN(c->m, n->m + n->low_cost);
    }
    if (n->offset_high != TERM0 && n->offset_high != TERM1) {
        NODE* c = n + n->offset_high;
        c->m = RMIN(c->m, n->m + n->high_cost);
    }
}

/* bdd_branch_instruction.h:171-198 */
static inline void FN(node_min_marginals)(const NODE* n, REAL mm[2])
{
    if (n->offset_low == TERM0 || n->offset_low == TERM1)
        mm[0] = n->m + n->low_cost;
    else
        mm[0] = n->m + n->low_cost + (n + n->offset_low)->m;
    if (n->offset_high == TERM0 || n->offset_high == TERM1)
        mm[1] = n->m + n->high_cost;
    else
        mm[1] = n->m + n->high_cost + (n + n->offset_high)->m;
}

void FN(oracle_destroy)(ORACLE* o)
{
    if (!o) return;
    free(o->nodes); free(o->bdd_layer_ptr); free(o->layer_node_ptr); free(o->layer_var);
    free(o->nr_bdds_per_var); free(o->delta_in); free(o->delta_out); free(o->mm_last); free(o);
}

/* add_bdds, bdd_parallel_mma_base.cpp:75-170.  instr = flat bdd_collection storage. */
ORACLE* FN(oracle_create)(const uint64_t* instr /* [n][3] = lo,hi,index */, const uint64_t* delims, uint64_t n_bdds)
{
    oracle_default_schedule();
    ORACLE* o = (ORACLE*)calloc(1, sizeof(ORACLE));
    o->n_bdds = n_bdds;
    o->n_threads = 1;
    size_t total_nodes = 0, total_layers = 0, max_v = 0;
    for (size_t b = 0; b < n_bdds; ++b) {
        size_t prev = (size_t)-3;
        for (size_t i = delims[b]; i < delims[b + 1]; ++i) {
            const uint64_t idx = instr[3 * i + 2];
            if (idx == TOPSINK || idx == BOTSINK) continue;
            ++total_nodes;
            if (idx != prev) { ++total_layers; prev = idx; }
            if (idx + 1 > max_v) max_v = idx + 1;
        }
    }
    o->n_nodes = total_nodes; o->n_layers = total_layers; o->n_vars = max_v;
    o->nodes = (NODE*)calloc(total_nodes ? total_nodes : 1, sizeof(NODE));
    o->bdd_layer_ptr = (size_t*)calloc(n_bdds + 1, sizeof(size_t));
    o->layer_node_ptr = (size_t*)calloc(total_layers + 1, sizeof(size_t));
    o->layer_var = (size_t*)calloc(total_layers ? total_layers : 1, sizeof(size_t));
    o->nr_bdds_per_var = (size_t*)calloc(max_v ? max_v : 1, sizeof(size_t));
    o->mm_last = (REAL*)calloc(total_layers ? total_layers : 1, sizeof(REAL));
    /* first node / first layer of every BDD (sequential, read-only over instr), then the fill in parallel over BDDs: the big arrays are
     * first touched by the threads that will work on them, so their pages spread over the host's NUMA nodes instead of all landing on
     * the node of the one thread that built the oracle (64 threads gave 2.9 x one thread on a 2-socket box). */
    size_t* node_first = (size_t*)malloc((n_bdds + 1) * sizeof(size_t));
    {
        size_t nn0 = 0, nl0 = 0;
        for (size_t b = 0; b < n_bdds; ++b) {
            node_first[b] = nn0;
            o->bdd_layer_ptr[b] = nl0;
            size_t prev = (size_t)-3;
            for (size_t i = delims[b]; i < delims[b + 1]; ++i) {
                const uint64_t idx = instr[3 * i + 2];
                if (idx == TOPSINK || idx == BOTSINK) continue;
                ++nn0;
                if (idx != prev) { ++nl0; prev = idx; o->nr_bdds_per_var[idx]++; }
            }
        }
        node_first[n_bdds] = nn0;
    }
    #pragma omp parallel for schedule(static, 512)
    for (size_t b = 0; b < n_bdds; ++b) {
        size_t nn = node_first[b], nl = o->bdd_layer_ptr[b];
        size_t prev = (size_t)-3;
        for (size_t i = delims[b]; i < delims[b + 1]; ++i) {
            const uint64_t lo = instr[3 * i + 0], hi = instr[3 * i + 1], idx = instr[3 * i + 2];
            if (idx == TOPSINK || idx == BOTSINK) continue;
            /* terminals are the last two entries of a BDD: non-terminal nodes keep their order */
            NODE* n = &o->nodes[nn];
            n->m = RINF; n->low_cost = 0; n->high_cost = 0;
            const uint64_t lo_idx = instr[3 * lo + 2], hi_idx = instr[3 * hi + 2];
            if (lo_idx == BOTSINK) n->offset_low = TERM0;
            else if (lo_idx == TOPSINK) n->offset_low = TERM1;
            else n->offset_low = (uint32_t)(lo - i);
            if (hi_idx == BOTSINK) n->offset_high = TERM0;
            else if (hi_idx == TOPSINK) n->offset_high = TERM1;
            else n->offset_high = (uint32_t)(hi - i);
            if (n->offset_low == TERM0) n->low_cost = RINF;    /* :137-138 */
            if (n->offset_high == TERM0) n->high_cost = RINF;  /* :140-141 */
            if (idx != prev) {
                o->layer_node_ptr[nl] = nn;
                o->layer_var[nl] = idx;
                o->mm_last[nl] = 0;
                ++nl; prev = idx;
            }
            ++nn;
        }
    }
    const size_t nn = node_first[n_bdds], nl = total_layers;
    free(node_first);
    o->bdd_layer_ptr[n_bdds] = nl;
    o->layer_node_ptr[nl] = nn;
    o->mp_state = MP_NONE;
    o->lb_valid = 0;
    o->lower_bound = -INFINITY;
    return o;
}

void FN(oracle_set_num_threads)(ORACLE* o, int n) { o->n_threads = n > 0 ? n : 1; }
uint64_t FN(oracle_nr_variables)(const ORACLE* o) { return o->n_vars; }
uint64_t FN(oracle_nr_bdds)(const ORACLE* o) { return o->n_bdds; }
uint64_t FN(oracle_nr_layers)(const ORACLE* o) { return o->n_layers; }  /* :1398-1402 */
uint64_t FN(oracle_nr_nodes)(const ORACLE* o) { return o->n_nodes; }
void FN(oracle_layer_info)(const ORACLE* o, int64_t* var, int64_t* bdd)
{
    for (size_t b = 0; b < o->n_bdds; ++b)
        for (size_t l = o->bdd_layer_ptr[b]; l < o->bdd_layer_ptr[b + 1]; ++l) {
            var[l] = (int64_t)o->layer_var[l];
            bdd[l] = (int64_t)b;
        }
}
void FN(oracle_nr_bdds_per_var)(const ORACLE* o, int64_t* out)
{
    for (size_t v = 0; v < o->n_vars; ++v) out[v] = (int64_t)o->nr_bdds_per_var[v];
}

/* update_costs, bdd_parallel_mma_base.cpp:626-696: cost/double(nr_bdds(var)) added to every arc
 * of the variable's layers that does not point to the bot sink. */
void FN(oracle_update_costs)(ORACLE* o, const double* lo, uint64_t n_lo, const double* hi, uint64_t n_hi)
{
    o->mp_state = MP_NONE;
    o->lb_valid = 0;
    for (size_t l = 0; l < o->n_layers; ++l) {
        const size_t var = o->layer_var[l];
        double lo_cost = 0.0, hi_cost = 0.0;
        if (o->nr_bdds_per_var[var] != 0) {
            if (var < n_lo) lo_cost = lo[var] / (double)o->nr_bdds_per_var[var];
            if (var < n_hi) hi_cost = hi[var] / (double)o->nr_bdds_per_var[var];
        }
        for (size_t i = o->layer_node_ptr[l]; i < o->layer_node_ptr[l + 1]; ++i) {
            /* `low_cost += lo_cost` with a double right-hand side (:674-677): the sum is formed in double and rounded once */
            if (o->nodes[i].offset_low != TERM0) o->nodes[i].low_cost = (REAL)((double)o->nodes[i].low_cost + lo_cost);
            if (o->nodes[i].offset_high != TERM0) o->nodes[i].high_cost = (REAL)((double)o->nodes[i].high_cost + hi_cost);
        }
    }
}

/* Per-layer arc costs (as net_solver_costs reads them, :1287-1297): value of an arc not into the bot sink. */
void FN(oracle_get_costs)(const ORACLE* o, REAL* lo, REAL* hi)
{
    for (size_t l = 0; l < o->n_layers; ++l) {
        REAL lc = -RINF, hc = -RINF;
        for (size_t i = o->layer_node_ptr[l]; i < o->layer_node_ptr[l + 1]; ++i) {
            if (o->nodes[i].offset_low != TERM0) lc = o->nodes[i].low_cost;
            if (o->nodes[i].offset_high != TERM0) hc = o->nodes[i].high_cost;
        }
        lo[l] = lc; hi[l] = hc;
    }
}

/* backward_run, :327-347 */
void FN(oracle_backward_run)(ORACLE* o)
{
    if (o->mp_state == MP_BWD) return;
    o->mp_state = MP_NONE;
    #pragma omp parallel for schedule(static, 512) num_threads(o->n_threads)
    for (ptrdiff_t b = (ptrdiff_t)o->n_bdds - 1; b >= 0; --b) {
        const size_t first = o->layer_node_ptr[o->bdd_layer_ptr[b]];
        const size_t last = o->layer_node_ptr[o->bdd_layer_ptr[b + 1]];
        for (ptrdiff_t i = (ptrdiff_t)last - 1; i >= (ptrdiff_t)first; --i) FN(backward_step)(&o->nodes[i]);
    }
    o->mp_state = MP_BWD;
}

/* forward_run, :299-325 */
void FN(oracle_forward_run)(ORACLE* o)
{
    if (o->mp_state == MP_FWD) return;
    o->mp_state = MP_NONE;
    #pragma omp parallel for schedule(static, 512) num_threads(o->n_threads)
    for (size_t b = 0; b < o->n_bdds; ++b) {
        const size_t first = o->layer_node_ptr[o->bdd_layer_ptr[b]];
        const size_t last = o->layer_node_ptr[o->bdd_layer_ptr[b + 1]];
        o->nodes[first].m = 0.0;
        for (size_t i = first; i < last; ++i) FN(prepare_forward_step)(&o->nodes[i]);
        for (size_t i = first; i < last; ++i) FN(forward_step)(&o->nodes[i]);
    }
    o->mp_state = MP_FWD;
}

/* compute_lower_bound*, :172-241 */
double FN(oracle_lower_bound)(ORACLE* o)
{
    if (o->lb_valid) return o->lower_bound;
    if (o->mp_state == MP_NONE) FN(oracle_backward_run)(o);
    double lb = 0.0;
    if (o->mp_state == MP_BWD) {
        for (size_t b = 0; b < o->n_bdds; ++b) lb += o->nodes[o->layer_node_ptr[o->bdd_layer_ptr[b]]].m;
    } else {
        for (size_t b = 0; b < o->n_bdds; ++b) {
            const size_t l = o->bdd_layer_ptr[b + 1] - 1;
            REAL bdd_lb = RINF;
            for (size_t i = o->layer_node_ptr[l]; i < o->layer_node_ptr[l + 1]; ++i) {
                REAL mm[2];
                FN(node_min_marginals)(&o->nodes[i], mm);
                bdd_lb = RMIN(bdd_lb, RMIN(mm[0], mm[1]));
            }
            lb += bdd_lb;
        }
    }
    o->lower_bound = lb;
    o->lb_valid = 1;
    return lb;
}

/* lower_bound_per_bdd_after_backward_pass, :283-297 */
void FN(oracle_lower_bound_per_bdd)(ORACLE* o, REAL* out)
{
    FN(oracle_backward_run)(o);
    for (size_t b = 0; b < o->n_bdds; ++b) out[b] = o->nodes[o->layer_node_ptr[o->bdd_layer_ptr[b]]].m;
}

/* atomic_add / atomic_store, bdd_parallel_mma_base.h:144-160 */
static inline void FN(atomic_add)(REAL* f, REAL d)
{
    if (d == 0) return;
    #pragma omp atomic
    *f += d;
}
static inline void FN(atomic_store)(REAL* f, REAL d)
{
    #pragma omp atomic write
    *f = d;
}

/* layer update shared by forward_mm(bdd) :832-872 and backward_mm(bdd) :904-941 */
static inline void FN(mm_layer_update)(ORACLE* o, size_t layer, size_t first, size_t last, size_t var, REAL omega, REAL* delta_out,
                                      int reverse)
{
    REAL cur_mm[2] = {RINF, RINF};
    if (!reverse) {
        for (size_t i = first; i < last; ++i) {
            REAL mm[2];
            FN(node_min_marginals)(&o->nodes[i], mm);
            cur_mm[0] = RMIN(mm[0], cur_mm[0]);
            cur_mm[1] = RMIN(mm[1], cur_mm[1]);
        }
    } else {
        for (ptrdiff_t i = (ptrdiff_t)last - 1; i >= (ptrdiff_t)first; --i) {
            REAL mm[2];
            FN(node_min_marginals)(&o->nodes[i], mm);
            cur_mm[0] = RMIN(mm[0], cur_mm[0]);
            cur_mm[1] = RMIN(mm[1], cur_mm[1]);
        }
    }
    const int f0 = isfinite(cur_mm[0]), f1 = isfinite(cur_mm[1]);
    /* the GPU solver keeps this amount per layer (mm_diff_out, bdd_cuda_parallel_mma.cu:36-39,131-137): omega * (m1 - m0),
     * 0 if either is non-finite; the CPU solver only accumulates it into delta_out.  Same value, same rounding. */
    o->mm_last[layer] = (f0 && f1) ? (cur_mm[0] < cur_mm[1] ? omega * (cur_mm[1] - cur_mm[0]) : -(omega * (cur_mm[0] - cur_mm[1]))) : (REAL)0;
    if (!f0) FN(atomic_store)(&delta_out[2 * var + 0], RINF);
    if (!f1) FN(atomic_store)(&delta_out[2 * var + 1], RINF);
    if (f0 && f1) {
        if (cur_mm[0] < cur_mm[1])
            FN(atomic_add)(&delta_out[2 * var + 1], omega * (cur_mm[1] - cur_mm[0]));
        else
            FN(atomic_add)(&delta_out[2 * var + 0], omega * (cur_mm[0] - cur_mm[1]));
    }
    for (size_t i = first; i < last; ++i) {
        NODE* n = &o->nodes[i];
        if (!f0) n->low_cost = RINF;
        if (!f1) n->high_cost = RINF;
        if (f0 && f1) {
            if (cur_mm[0] < cur_mm[1])
                n->high_cost += omega * (cur_mm[0] - cur_mm[1]);
            else
                n->low_cost += omega * (cur_mm[1] - cur_mm[0]);
        }
    }
}

/* forward_mm(bdd_nr, omega, delta_out, delta_in), :814-889 */
static void FN(forward_mm_bdd)(ORACLE* o, size_t b, REAL omega, REAL* delta_out, const REAL* delta_in)
{
    o->nodes[o->layer_node_ptr[o->bdd_layer_ptr[b]]].m = 0.0;
    for (size_t l = o->bdd_layer_ptr[b]; l < o->bdd_layer_ptr[b + 1]; ++l) {
        const size_t first = o->layer_node_ptr[l], last = o->layer_node_ptr[l + 1];
        const size_t var = o->layer_var[l];
        FN(mm_layer_update)(o, l, first, last, var, omega, delta_out, 0);
        if (l + 1 < o->bdd_layer_ptr[b + 1])
            for (size_t i = o->layer_node_ptr[l + 1]; i < o->layer_node_ptr[l + 2]; ++i) o->nodes[i].m = RINF;
        for (size_t i = first; i < last; ++i) {
            o->nodes[i].low_cost += delta_in[2 * var + 0];
            o->nodes[i].high_cost += delta_in[2 * var + 1];
            FN(forward_step)(&o->nodes[i]);
        }
    }
}

/* backward_mm(bdd_nr, omega, delta_out, delta_in), :891-956 */
static REAL FN(backward_mm_bdd)(ORACLE* o, size_t b, REAL omega, REAL* delta_out, const REAL* delta_in)
{
    for (ptrdiff_t l = (ptrdiff_t)o->bdd_layer_ptr[b + 1] - 1; l >= (ptrdiff_t)o->bdd_layer_ptr[b]; --l) {
        const size_t first = o->layer_node_ptr[l], last = o->layer_node_ptr[l + 1];
        const size_t var = o->layer_var[l];
        FN(mm_layer_update)(o, (size_t)l, first, last, var, omega, delta_out, 1);
        for (ptrdiff_t i = (ptrdiff_t)last - 1; i >= (ptrdiff_t)first; --i) {
            o->nodes[i].low_cost += delta_in[2 * var + 0];
            o->nodes[i].high_cost += delta_in[2 * var + 1];
            FN(backward_step)(&o->nodes[i]);
        }
    }
    return o->nodes[o->layer_node_ptr[o->bdd_layer_ptr[b]]].m;
}

static void FN(ensure_delta_out)(ORACLE* o)
{
    if (!o->have_delta_out) {
        o->delta_out = (REAL*)calloc(2 * (o->n_vars ? o->n_vars : 1), sizeof(REAL));
        o->have_delta_out = 1;
    } else {
        memset(o->delta_out, 0, 2 * o->n_vars * sizeof(REAL));
    }
}

/* forward_mm(omega, delta), :958-982.  NB: the reference ignores `omega` here and uses 0.5 (:975). */
void FN(oracle_forward_mm)(ORACLE* o, REAL omega, REAL* delta /* 2*n_vars in/out */)
{
    (void)omega;
    FN(oracle_backward_run)(o);  /* forward_mm(bdd) calls backward_run() first, :819 */
    FN(ensure_delta_out)(o);
    #pragma omp parallel for schedule(runtime) num_threads(o->n_threads)
    for (size_t b = 0; b < o->n_bdds; ++b) FN(forward_mm_bdd)(o, b, (REAL)0.5, o->delta_out, delta);
    /* std::swap(delta_out_, delta) */
    for (size_t i = 0; i < 2 * o->n_vars; ++i) { REAL t = o->delta_out[i]; o->delta_out[i] = delta[i]; delta[i] = t; }
    o->lb_valid = 0;
    o->mp_state = MP_FWD;
}

/* backward_mm(omega, delta), :984-1009 */
double FN(oracle_backward_mm)(ORACLE* o, REAL omega, REAL* delta)
{
    (void)omega;
    FN(ensure_delta_out)(o);
    double lb = 0.0;
    #pragma omp parallel for schedule(runtime) reduction(+ : lb) num_threads(o->n_threads)
    for (size_t b = 0; b < o->n_bdds; ++b) lb += FN(backward_mm_bdd)(o, b, (REAL)0.5, o->delta_out, delta);
    for (size_t i = 0; i < 2 * o->n_vars; ++i) { REAL t = o->delta_out[i]; o->delta_out[i] = delta[i]; delta[i] = t; }
    o->lb_valid = 0;
    o->mp_state = MP_BWD;
    return lb;
}

/* iteration(), :1011-1044 */
void FN(oracle_iteration)(ORACLE* o)
{
    FN(oracle_backward_run)(o);
    if (!o->have_delta_in) {
        o->delta_in = (REAL*)calloc(2 * (o->n_vars ? o->n_vars : 1), sizeof(REAL));
        o->have_delta_in = 1;
    }
    /* The swap in forward_mm/backward_mm is implemented by copying above; do it by pointer here. */
    for (int pass = 0; pass < 2; ++pass) {
        FN(ensure_delta_out)(o);
        if (pass == 0) {
            #pragma omp parallel for schedule(runtime) num_threads(o->n_threads)
            for (size_t b = 0; b < o->n_bdds; ++b) FN(forward_mm_bdd)(o, b, (REAL)0.5, o->delta_out, o->delta_in);
            o->mp_state = MP_FWD;
        } else {
            double lb = 0.0;
            #pragma omp parallel for schedule(runtime) reduction(+ : lb) num_threads(o->n_threads)
            for (size_t b = 0; b < o->n_bdds; ++b) lb += FN(backward_mm_bdd)(o, b, (REAL)0.5, o->delta_out, o->delta_in);
            o->lower_bound = lb;
            o->mp_state = MP_BWD;
        }
        REAL* t = o->delta_out; o->delta_out = o->delta_in; o->delta_in = t;
        /* average_mms, :1021-1033 */
        #pragma omp parallel for num_threads(o->n_threads)
        for (size_t v = 0; v < o->n_vars; ++v) {
            if (o->nr_bdds_per_var[v] > 0) {
                o->delta_in[2 * v + 0] /= (REAL)o->nr_bdds_per_var[v];
                o->delta_in[2 * v + 1] /= (REAL)o->nr_bdds_per_var[v];
            }
        }
    }
    o->lb_valid = 1;
}

/* distribute_delta(), :1046-1072 */
void FN(oracle_distribute_delta)(ORACLE* o)
{
    o->mp_state = MP_NONE;
    o->lb_valid = 0;
    if (!o->have_delta_in) return;
    for (size_t l = 0; l < o->n_layers; ++l) {
        const size_t var = o->layer_var[l];
        for (size_t i = o->layer_node_ptr[l]; i < o->layer_node_ptr[l + 1]; ++i) {
            o->nodes[i].low_cost += o->delta_in[2 * var + 0];
            o->nodes[i].high_cost += o->delta_in[2 * var + 1];
        }
    }
    memset(o->delta_in, 0, 2 * o->n_vars * sizeof(REAL));
}

void FN(oracle_get_delta_in)(const ORACLE* o, REAL* out)
{
    if (o->have_delta_in) memcpy(out, o->delta_in, 2 * o->n_vars * sizeof(REAL));
    else memset(out, 0, 2 * o->n_vars * sizeof(REAL));
}

/* min_marginals(), :372-416 — output in BDD-major layer order (the caller transposes, :1074-1094). */
void FN(oracle_min_marginals)(ORACLE* o, double* out /* [n_layers][2] */)
{
    FN(oracle_backward_run)(o);
    for (size_t b = 0; b < o->n_bdds; ++b) {
        o->nodes[o->layer_node_ptr[o->bdd_layer_ptr[b]]].m = 0.0;
        for (size_t l = o->bdd_layer_ptr[b]; l < o->bdd_layer_ptr[b + 1]; ++l) {
            REAL mm[2] = {RINF, RINF};
            const size_t first = o->layer_node_ptr[l], last = o->layer_node_ptr[l + 1];
            for (size_t i = first; i < last; ++i) {
                REAL cur[2];
                FN(node_min_marginals)(&o->nodes[i], cur);
                mm[0] = RMIN(mm[0], cur[0]);
                mm[1] = RMIN(mm[1], cur[1]);
            }
            out[2 * l + 0] = mm[0];
            out[2 * l + 1] = mm[1];
            for (size_t i = first; i < last; ++i) FN(prepare_forward_step)(&o->nodes[i]);
            for (size_t i = first; i < last; ++i) FN(forward_step)(&o->nodes[i]);
        }
    }
    o->mp_state = MP_FWD;
}

/* bdds_solution_vec(), :1197-1275 */
void FN(oracle_bdds_solution_vec)(ORACLE* o, char* sol /* n_layers */)
{
    FN(oracle_backward_run)(o);
    for (size_t b = 0; b < o->n_bdds; ++b) {
        const size_t root = o->layer_node_ptr[o->bdd_layer_ptr[b]];
        o->nodes[root].m = 0.0;
        size_t next_node = root;
        for (size_t l = o->bdd_layer_ptr[b]; l < o->bdd_layer_ptr[b + 1]; ++l) {
            const size_t first = o->layer_node_ptr[l], last = o->layer_node_ptr[l + 1];
            for (size_t i = first; i < last; ++i) {
                if (next_node == i) {
                    REAL cur[2];
                    FN(node_min_marginals)(&o->nodes[i], cur);
                    const NODE* n = &o->nodes[i];
                    if (cur[0] < cur[1]) {
                        sol[l] = 0;
                        if (n->offset_low != TERM0 && n->offset_low != TERM1) next_node = i + n->offset_low;
                    } else {
                        sol[l] = 1;
                        if (n->offset_high != TERM0 && n->offset_high != TERM1) next_node = i + n->offset_high;
                    }
                    break;
                }
            }
            for (size_t i = first; i < last; ++i) FN(prepare_forward_step)(&o->nodes[i]);
            for (size_t i = first; i < last; ++i) FN(forward_step)(&o->nodes[i]);
        }
    }
    o->mp_state = MP_FWD;
}

/* make_dual_feasible, :1346-1367 */
void FN(oracle_make_dual_feasible)(const ORACLE* o, REAL* duals)
{
    REAL* sum = (REAL*)calloc(o->n_vars ? o->n_vars : 1, sizeof(REAL));
    for (size_t l = 0; l < o->n_layers; ++l) sum[o->layer_var[l]] += duals[l];
    for (size_t v = 0; v < o->n_vars; ++v) sum[v] /= (REAL)o->nr_bdds_per_var[v];
    for (size_t l = 0; l < o->n_layers; ++l) duals[l] -= sum[o->layer_var[l]];
    free(sum);
}

/* gradient_step, :1369-1396 */
void FN(oracle_gradient_step)(ORACLE* o, const REAL* duals, double step_size)
{
    o->mp_state = MP_NONE;
    o->lb_valid = 0;
    for (size_t l = 0; l < o->n_layers; ++l)
        for (size_t i = o->layer_node_ptr[l]; i < o->layer_node_ptr[l + 1]; ++i)
            if (o->nodes[i].offset_high != TERM0) o->nodes[i].high_cost += step_size * duals[l];
}

/* net_solver_costs of the GPU solver (compute_net_costs_func, bdd_cuda_parallel_mma.cu:432-463): hi - lo + the layer's own
 * deferred min-marginal difference.  (The CPU solver's net_solver_costs, bdd_parallel_mma_base.cpp:1284-1326, spreads the
 * deferred amount evenly over the BDDs of the variable instead; the L-BFGS oracle follows the GPU definition, which is the
 * path under test.)  Layer costs are read as oracle_get_costs does. */
void FN(oracle_net_solver_costs)(const ORACLE* o, REAL* out)
{
    for (size_t l = 0; l < o->n_layers; ++l) {
        REAL lo = 0, hi = 0;
        for (size_t i = o->layer_node_ptr[l]; i < o->layer_node_ptr[l + 1]; ++i) {
            if (o->nodes[i].offset_low != TERM0) lo = o->nodes[i].low_cost;
            if (o->nodes[i].offset_high != TERM0) hi = o->nodes[i].high_cost;
        }
        out[l] = hi - lo + o->mm_last[l];
    }
}
void FN(oracle_get_mm_last)(const ORACLE* o, REAL* out) { memcpy(out, o->mm_last, o->n_layers * sizeof(REAL)); }

#undef CAT_
#undef CAT
#undef FN
#undef NODE
#undef ORACLE
#undef MP_NONE
#undef MP_FWD
#undef MP_BWD

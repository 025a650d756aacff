/*
 * cuda_rule_oracle_impl.h — body of the "GPU-rule" oracle, included once per precision by cuda_rule_oracle.c
 * with REAL, SUFFIX and RINF defined.
 *
 * TEST INFRASTRUCTURE ONLY (see cuda_rule_oracle.c header).
 *
 * Restates, BDD by BDD and layer by layer, what the reference's GPU solver bdd_cuda_parallel_mma<REAL> computes per pass
 * (SURVEY.md §8 a'), with omega as an argument and with the GPU's rule for non-finite min-marginals.  Per node: F (cost_from_root_),
 * T (cost_from_terminal_); per layer: lo / hi (lo_cost_, hi_cost_: finite always, arcs into the bot sink get their +inf from
 * T[bot] = +inf, bdd_cuda_base.cu:217-227), mm (deffered_mm_diff_).  Layers are in BDD-major order (the CPU oracle's order).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)
#define CR CAT(cuda_rule, SUFFIX)

#define CR_TOP (-1) /* TOP_SINK_INDICATOR_CUDA */
#define CR_BOT (-2) /* BOT_SINK_INDICATOR_CUDA */

typedef struct CR {
    size_t n_bdds, n_vars, n_nodes, n_layers; /* non-terminal nodes and layers */
    int64_t* lo_child;                         /* per node: index of the child node, CR_TOP or CR_BOT */
    int64_t* hi_child;
    size_t* node_layer;
    size_t* bdd_layer_ptr;  /* n_bdds + 1 */
    size_t* layer_node_ptr; /* n_layers + 1 */
    size_t* layer_var;
    size_t* nr_bdds_per_var;
    REAL *lo_cost, *hi_cost, *lo_out, *hi_out, *mm; /* per layer */
    REAL *F, *T;                                    /* per node */
    REAL* delta;                                    /* delta_lo_hi_, 2 * n_vars, created by the first iteration() */
    int fwd_valid, bwd_valid;
} CR;

void FN(cr_destroy)(CR* o)
{
    if (!o) return;
    free(o->lo_child); free(o->hi_child); free(o->node_layer); free(o->bdd_layer_ptr); free(o->layer_node_ptr); free(o->layer_var);
    free(o->nr_bdds_per_var); free(o->lo_cost); free(o->hi_cost); free(o->lo_out); free(o->hi_out); free(o->mm); free(o->F); free(o->T);
    free(o->delta); free(o);
}

/* bdd_cuda_base ctor, bdd_cuda_base.cu:31-46,55-144: flat bdd_collection storage -> nodes, children, layers; all arc costs 0 */
CR* FN(cr_create)(const uint64_t* instr /* [n][3] = lo, hi, index */, const uint64_t* delims, uint64_t n_bdds)
{
    CR* o = (CR*)calloc(1, sizeof(CR));
    o->n_bdds = n_bdds;
    size_t nn = 0, nl = 0, nv = 0;
    for (size_t b = 0; b < n_bdds; ++b) {
        uint64_t prev = CR_TOPSINK;
        for (size_t i = delims[b]; i < delims[b + 1]; ++i) {
            const uint64_t idx = instr[3 * i + 2];
            if (idx == CR_TOPSINK || idx == CR_BOTSINK) continue;
            ++nn;
            if (idx != prev) { ++nl; prev = idx; }
            if (idx + 1 > nv) nv = idx + 1;
        }
    }
    o->n_nodes = nn; o->n_layers = nl; o->n_vars = nv;
    o->lo_child = (int64_t*)malloc((nn ? nn : 1) * sizeof(int64_t));
    o->hi_child = (int64_t*)malloc((nn ? nn : 1) * sizeof(int64_t));
    o->node_layer = (size_t*)malloc((nn ? nn : 1) * sizeof(size_t));
    o->bdd_layer_ptr = (size_t*)calloc(n_bdds + 1, sizeof(size_t));
    o->layer_node_ptr = (size_t*)calloc(nl + 1, sizeof(size_t));
    o->layer_var = (size_t*)calloc(nl ? nl : 1, sizeof(size_t));
    o->nr_bdds_per_var = (size_t*)calloc(nv ? nv : 1, sizeof(size_t));
    o->lo_cost = (REAL*)calloc(nl ? nl : 1, sizeof(REAL));
    o->hi_cost = (REAL*)calloc(nl ? nl : 1, sizeof(REAL));
    o->lo_out = (REAL*)calloc(nl ? nl : 1, sizeof(REAL));
    o->hi_out = (REAL*)calloc(nl ? nl : 1, sizeof(REAL));
    o->mm = (REAL*)calloc(nl ? nl : 1, sizeof(REAL));
    o->F = (REAL*)calloc(nn ? nn : 1, sizeof(REAL));
    o->T = (REAL*)calloc(nn ? nn : 1, sizeof(REAL));
    size_t node = 0, layer = 0;
    for (size_t b = 0; b < n_bdds; ++b) {
        const size_t d0 = delims[b], d1 = delims[b + 1];
        /* flat index -> node index of this BDD's non-terminal instructions */
        int64_t* map = (int64_t*)malloc((d1 - d0 ? d1 - d0 : 1) * sizeof(int64_t));
        size_t k = node;
        for (size_t i = d0; i < d1; ++i) {
            const uint64_t idx = instr[3 * i + 2];
            map[i - d0] = idx == CR_TOPSINK ? CR_TOP : (idx == CR_BOTSINK ? CR_BOT : (int64_t)k++);
        }
        o->bdd_layer_ptr[b] = layer;
        uint64_t prev = CR_TOPSINK;
        for (size_t i = d0; i < d1; ++i) {
            const uint64_t idx = instr[3 * i + 2];
            if (idx == CR_TOPSINK || idx == CR_BOTSINK) continue;
            if (idx != prev) {
                o->layer_node_ptr[layer] = node;
                o->layer_var[layer] = (size_t)idx;
                ++o->nr_bdds_per_var[idx];
                ++layer;
                prev = idx;
            }
            o->node_layer[node] = layer - 1;
            o->lo_child[node] = map[instr[3 * i + 0] - d0];
            o->hi_child[node] = map[instr[3 * i + 1] - d0];
            ++node;
        }
        free(map);
    }
    o->bdd_layer_ptr[n_bdds] = layer;
    o->layer_node_ptr[layer] = node;
    return o;
}

uint64_t FN(cr_nr_variables)(const CR* o) { return o->n_vars; }
uint64_t FN(cr_nr_layers)(const CR* o) { return o->n_layers; }
uint64_t FN(cr_nr_bdds)(const CR* o) { return o->n_bdds; }

/* update_costs, bdd_cuda_base.cu:439-503 (set_vars_costs_func :455-474): cost[layer] += c[var] / nr_bdds(var); a side whose vector is
 * non-empty but shorter than nr_variables() is SET to 0 on the tail variables (:465-469).  The quotient and the sum are formed in double and
 * rounded once — the build's documented deviation (as the CPU solver, bdd_parallel_mma_base.cpp:640,651). */
void FN(cr_update_costs)(CR* o, const double* lo, uint64_t n_lo, const double* hi, uint64_t n_hi)
{
    for (size_t l = 0; l < o->n_layers; ++l) {
        const size_t v = o->layer_var[l];
        const double nb = (double)o->nr_bdds_per_var[v];
        if (n_lo) o->lo_cost[l] = v < n_lo ? (REAL)((double)o->lo_cost[l] + lo[v] / nb) : (REAL)0;
        if (n_hi) o->hi_cost[l] = v < n_hi ? (REAL)((double)o->hi_cost[l] + hi[v] / nb) : (REAL)0;
    }
    o->fwd_valid = o->bwd_valid = 0;
}

void FN(cr_set_layer_costs)(CR* o, const REAL* lo, const REAL* hi)
{
    memcpy(o->lo_cost, lo, o->n_layers * sizeof(REAL));
    memcpy(o->hi_cost, hi, o->n_layers * sizeof(REAL));
    o->fwd_valid = o->bwd_valid = 0;
}
void FN(cr_get_layer_costs)(const CR* o, REAL* lo, REAL* hi)
{
    memcpy(lo, o->lo_cost, o->n_layers * sizeof(REAL));
    memcpy(hi, o->hi_cost, o->n_layers * sizeof(REAL));
}
void FN(cr_get_mm)(const CR* o, REAL* out) { memcpy(out, o->mm, o->n_layers * sizeof(REAL)); }
void FN(cr_layer_info)(const CR* o, int64_t* var, int64_t* bdd)
{
    for (size_t b = 0; b < o->n_bdds; ++b)
        for (size_t l = o->bdd_layer_ptr[b]; l < o->bdd_layer_ptr[b + 1]; ++l) { var[l] = (int64_t)o->layer_var[l]; bdd[l] = (int64_t)b; }
}

/* cost_from_terminal_ of a child: set_special_nodes_costs, bdd_cuda_base.cu:217-227 */
static inline REAL FN(cr_T)(const CR* o, int64_t c) { return c == CR_TOP ? (REAL)0 : (c == CR_BOT ? RINF : o->T[c]); }
static inline REAL FN(cr_min)(REAL a, REAL b) { return b < a ? b : a; } /* CUDA min(): no NaN arises here */

/* backward_run(false) / backward_step, bdd_cuda_base.cu:646-667,669-713 */
void FN(cr_backward_run)(CR* o)
{
    if (o->bwd_valid) return;
    for (size_t l = o->n_layers; l-- > 0;)
        for (size_t u = o->layer_node_ptr[l]; u < o->layer_node_ptr[l + 1]; ++u)
            o->T[u] = FN(cr_min)(FN(cr_T)(o, o->hi_child[u]) + o->hi_cost[l], FN(cr_T)(o, o->lo_child[u]) + o->lo_cost[l]);
    o->bwd_valid = 1;
}

/* flush_costs_from_root, bdd_cuda_base.cu:1438-1445 */
static void FN(cr_flush_from_root)(CR* o)
{
    for (size_t u = 0; u < o->n_nodes; ++u) o->F[u] = RINF;
    for (size_t b = 0; b < o->n_bdds; ++b)
        if (o->bdd_layer_ptr[b] < o->bdd_layer_ptr[b + 1]) o->F[o->layer_node_ptr[o->bdd_layer_ptr[b]]] = (REAL)0;
}

/* forward_run / forward_step, bdd_cuda_base.cu:560-612 */
void FN(cr_forward_run)(CR* o)
{
    if (o->fwd_valid) return;
    FN(cr_flush_from_root)(o);
    for (size_t l = 0; l < o->n_layers; ++l)
        for (size_t u = o->layer_node_ptr[l]; u < o->layer_node_ptr[l + 1]; ++u) {
            const int64_t lc = o->lo_child[u], hc = o->hi_child[u];
            if (lc >= 0) o->F[lc] = FN(cr_min)(o->F[lc], o->F[u] + o->lo_cost[l]);
            if (hc >= 0) o->F[hc] = FN(cr_min)(o->F[hc], o->F[u] + o->hi_cost[l]);
        }
    o->fwd_valid = 1;
}

/* lower_bound, bdd_cuda_base.cu:1243-1251: backward_run(false); sum of the roots' costs from terminal, in double */
double FN(cr_lower_bound)(CR* o)
{
    FN(cr_backward_run)(o);
    double lb = 0.0;
    for (size_t b = 0; b < o->n_bdds; ++b)
        if (o->bdd_layer_ptr[b] < o->bdd_layer_ptr[b + 1]) lb += (double)o->T[o->layer_node_ptr[o->bdd_layer_ptr[b]]];
    return lb;
}

/* min_marginals_from_directional_costs, bdd_cuda_parallel_mma.cu:59-139, of one layer: the per-node sums are formed left to right (:83-84),
 * the difference is omega * (mm_hi - mm_lo), or 0 unless BOTH minima are finite (compute_mm_diff_flush_mm_lo, :29-42) */
static inline REAL FN(cr_layer_mm)(const CR* o, size_t l, REAL omega)
{
    REAL m0 = RINF, m1 = RINF;
    for (size_t u = o->layer_node_ptr[l]; u < o->layer_node_ptr[l + 1]; ++u) {
        m0 = FN(cr_min)(m0, (o->F[u] + o->lo_cost[l]) + FN(cr_T)(o, o->lo_child[u]));
        m1 = FN(cr_min)(m1, (o->F[u] + o->hi_cost[l]) + FN(cr_T)(o, o->hi_child[u]));
    }
    if (!isfinite(m1) || !isfinite(m0)) return (REAL)0;
    return omega * (m1 - m0);
}

/* compute_delta, bdd_cuda_parallel_mma.cu:358-393 (the order of the atomicAdds is the one freedom of the GPU implementation) */
static void FN(cr_compute_delta)(const CR* o, REAL* delta)
{
    memset(delta, 0, 2 * o->n_vars * sizeof(REAL));
    for (size_t l = 0; l < o->n_layers; ++l) {
        const REAL m = o->mm[l];
        if (m > 0) delta[2 * o->layer_var[l] + 1] += m;
        else if (m < 0) delta[2 * o->layer_var[l]] += -m;
    }
}

/* forward_mm(omega, delta_lo_hi), bdd_cuda_parallel_mma.cu:164-257 */
void FN(cr_forward_mm)(CR* o, REAL omega, REAL* delta)
{
    if (!o->bwd_valid) FN(cr_backward_run)(o); /* :212-213 */
    FN(cr_flush_from_root)(o);
    for (size_t l = 0; l < o->n_layers; ++l) { /* BDDs are independent inside a pass: hop-major (reference) == BDD-major (here) */
        const size_t v = o->layer_var[l];
        const REAL mm = FN(cr_layer_mm)(o, l, omega);
        o->mm[l] = mm;
        const REAL lo = (o->lo_cost[l] + FN(cr_min)(mm, (REAL)0)) + delta[2 * v];       /* :191 */
        const REAL hi = (o->hi_cost[l] + FN(cr_min)(-mm, (REAL)0)) + delta[2 * v + 1];  /* :196 */
        for (size_t u = o->layer_node_ptr[l]; u < o->layer_node_ptr[l + 1]; ++u) {
            const int64_t lc = o->lo_child[u], hc = o->hi_child[u];
            if (lc >= 0) o->F[lc] = FN(cr_min)(o->F[lc], o->F[u] + lo); /* children that are terminals: their F is never read */
            if (hc >= 0) o->F[hc] = FN(cr_min)(o->F[hc], o->F[u] + hi);
        }
        o->lo_out[l] = lo;
        o->hi_out[l] = hi;
    }
    { REAL* t = o->lo_cost; o->lo_cost = o->lo_out; o->lo_out = t; }
    { REAL* t = o->hi_cost; o->hi_cost = o->hi_out; o->hi_out = t; }
    FN(cr_compute_delta)(o, delta);
    o->fwd_valid = 1;
    o->bwd_valid = 0;
}

/* backward_mm(omega, delta_lo_hi), bdd_cuda_parallel_mma.cu:259-346; returns 0 on success, -1 if the forward state is not valid (:304) */
int FN(cr_backward_mm)(CR* o, REAL omega, REAL* delta)
{
    if (!o->fwd_valid) return -1;
    for (size_t l = o->n_layers; l-- > 0;) {
        const size_t v = o->layer_var[l];
        const REAL mm = FN(cr_layer_mm)(o, l, omega);
        o->mm[l] = mm;
        const REAL hi = (o->hi_cost[l] + FN(cr_min)(-mm, (REAL)0)) + delta[2 * v + 1]; /* :286 */
        const REAL lo = (o->lo_cost[l] + FN(cr_min)(mm, (REAL)0)) + delta[2 * v];      /* :287 */
        for (size_t u = o->layer_node_ptr[l]; u < o->layer_node_ptr[l + 1]; ++u)
            o->T[u] = FN(cr_min)(hi + FN(cr_T)(o, o->hi_child[u]), lo + FN(cr_T)(o, o->lo_child[u])); /* :292 */
        o->lo_out[l] = lo;
        o->hi_out[l] = hi;
    }
    { REAL* t = o->lo_cost; o->lo_cost = o->lo_out; o->lo_out = t; }
    { REAL* t = o->hi_cost; o->hi_cost = o->hi_out; o->hi_out = t; }
    FN(cr_compute_delta)(o, delta);
    o->fwd_valid = 0;
    o->bwd_valid = 1;
    return 0;
}

/* normalize_delta, bdd_cuda_parallel_mma.cu:410-430 (0 / 0 of a variable in no BDD is never read; left at 0 here) */
void FN(cr_normalize_delta)(const CR* o, REAL* delta)
{
    for (size_t v = 0; v < o->n_vars; ++v)
        if (o->nr_bdds_per_var[v] > 0) {
            delta[2 * v] /= (REAL)o->nr_bdds_per_var[v];
            delta[2 * v + 1] /= (REAL)o->nr_bdds_per_var[v];
        }
}

/* iteration(omega), bdd_cuda_parallel_mma.cu:142-153 */
void FN(cr_iteration)(CR* o, REAL omega)
{
    if (!o->delta) o->delta = (REAL*)calloc(2 * (o->n_vars ? o->n_vars : 1), sizeof(REAL));
    FN(cr_forward_mm)(o, omega, o->delta);
    FN(cr_normalize_delta)(o, o->delta);
    FN(cr_backward_mm)(o, omega, o->delta);
    FN(cr_normalize_delta)(o, o->delta);
}
void FN(cr_get_delta)(const CR* o, REAL* out)
{
    if (o->delta) memcpy(out, o->delta, 2 * o->n_vars * sizeof(REAL));
    else memset(out, 0, 2 * o->n_vars * sizeof(REAL));
}

/* distribute_delta, bdd_cuda_base.cu:1396-1436 */
void FN(cr_distribute_delta)(CR* o)
{
    for (size_t l = 0; l < o->n_layers; ++l) {
        const REAL m = o->mm[l];
        if (m > 0) o->hi_cost[l] += m;
        else o->lo_cost[l] -= m;
        o->mm[l] = (REAL)0;
    }
    if (o->delta) memset(o->delta, 0, 2 * o->n_vars * sizeof(REAL));
    o->fwd_valid = o->bwd_valid = 0;
}

/* min_marginals_cuda(false), bdd_cuda_base.cu:716-786: forward_run; backward_run(true); per layer the minima of the nodes' path costs
 * hi_path = F + (T[hi] + hi), lo_path = F + (T[lo] + lo) (backward_step_with_path_costs, :633-641) */
void FN(cr_min_marginals)(CR* o, REAL* mm0, REAL* mm1)
{
    FN(cr_forward_run)(o);
    o->bwd_valid = 0;
    FN(cr_backward_run)(o);
    for (size_t l = 0; l < o->n_layers; ++l) {
        REAL a = RINF, b = RINF;
        for (size_t u = o->layer_node_ptr[l]; u < o->layer_node_ptr[l + 1]; ++u) {
            a = FN(cr_min)(a, o->F[u] + (FN(cr_T)(o, o->lo_child[u]) + o->lo_cost[l]));
            b = FN(cr_min)(b, o->F[u] + (FN(cr_T)(o, o->hi_child[u]) + o->hi_cost[l]));
        }
        mm0[l] = a;
        mm1[l] = b;
    }
}

#undef CAT_
#undef CAT
#undef FN
#undef CR
#undef CR_TOP
#undef CR_BOT

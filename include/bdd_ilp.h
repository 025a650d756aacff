/* bdd_ilp.h — C-ABI of the host-side input stage that feeds the hot path: .lp reader, ILP -> QBDD conversion,
 * long-BDD splitting.  Implemented in libbdd_mma_hip.so (bdd_amd/csrc/host/), pure host code: every entry point
 * works without a GPU.
 *
 * Replaces, for the path of SURVEY.md §8(f-2..4):
 *   ILP_parser::parse_file / parse_string            src/ILP/ILP_parser.cpp:24-140, :380-420
 *   bdd_preprocessor::add_ilp (linear rows)          src/bdd_conversion/bdd_preprocessor.cpp:123-336
 *   bdd_collection::split_qbdd + the splitting stage src/bdd_collection/bdd_collection.cpp:507-949,
 *                                                    src/bdd_conversion/bdd_preprocessor.cpp:372-415
 * The output of bddilp_to_bdds is exactly what bddmma_create (bdd_mma.h) consumes.
 * Every call returns 0 or a negative code; bddilp_last_error() gives the message (thread-local). */
#ifndef BDD_ILP_H
#define BDD_ILP_H

#include <stdint.h>

#include "bdd_mma.h"

#ifdef __cplusplus
extern "C" {
#endif

#define BDDILP_OK 0
#define BDDILP_ERR_PARSE (-1)
#define BDDILP_ERR_INFEASIBLE (-2)
#define BDDILP_ERR_INVALID_ARGUMENT (-3)

#define BDDILP_LE (-1)
#define BDDILP_EQ 0
#define BDDILP_GE 1

typedef struct bddilp bddilp;            /* LPMP::ILP_input */
typedef struct bddilp_bdds bddilp_bdds;  /* BDD::bdd_collection (flat storage) */

const char* bddilp_last_error(void);

/* ILP_parser::parse_string: variables are numbered by first appearance, objective first. */
int bddilp_parse_lp(const char* lp_text, bddilp** out);
/* OPB_parser::parse_string (src/ILP/OPB_parser.cpp:23-60, :240-252): `* comments`, `min: terms ;`, `terms rel int ;` rows. */
int bddilp_parse_opb(const char* opb_text, bddilp** out);
/* bdd_solver::read_ILP for strings (bdd_solver.cpp:59-63): the .lp grammar first, then OPB. */
int bddilp_parse(const char* text, bddilp** out);
void bddilp_destroy(bddilp* ilp);
uint64_t bddilp_nr_variables(const bddilp* ilp);
uint64_t bddilp_nr_constraints(const bddilp* ilp);
const char* bddilp_variable_name(const bddilp* ilp, uint64_t var);
/* objective[nr_variables] and the constant term */
int bddilp_objective(const bddilp* ilp, double* objective, double* constant);
/* row c: *n terms (coeffs / vars need room for bddilp_constraint_size(c) entries), relation, rhs, name */
uint64_t bddilp_constraint_size(const bddilp* ilp, uint64_t c);
int bddilp_constraint(const bddilp* ilp, uint64_t c, int64_t* coeffs, uint64_t* vars, int* ineq, int64_t* rhs);
const char* bddilp_constraint_name(const bddilp* ilp, uint64_t c);
/* ILP_input::constraint::normalize on every row ("normalize constraints") */
int bddilp_normalize(bddilp* ilp);

/* bdd_preprocessor::add_ilp: one QBDD per row (trivially true rows are skipped, an infeasible row fails with
 * BDDILP_ERR_INFEASIBLE).  split != 0: BDDs longer than split_length variables are cut by split_qbdd
 * (split_length 0: the reference's occupancy rule with MI355X's figures); split == 2 also adds split_qbdd's
 * implication BDD over the auxiliary variables (bdd_collection.cpp:801-941, JSON "split bdds": {"implication bdd": true}). */
int bddilp_to_bdds(const bddilp* ilp, int split, uint64_t split_length, bddilp_bdds** out);
/* single rows, for tests and front ends that build their own collections */
int bddilp_bdds_create(bddilp_bdds** out);
int bddilp_bdds_add_row(bddilp_bdds* b, const int64_t* coeffs, const uint64_t* vars, uint64_t n, int ineq, int64_t rhs,
                        int* status /* 0 added, 1 trivially true (skipped), 2 infeasible */);
int bddilp_bdds_split(bddilp_bdds* b, uint64_t nr_variables, uint64_t split_length, int with_implication_bdd, uint64_t* nr_split,
                      uint64_t* nr_variables_after);
void bddilp_bdds_destroy(bddilp_bdds* b);
uint64_t bddilp_bdds_nr_bdds(const bddilp_bdds* b);
uint64_t bddilp_bdds_nr_instructions(const bddilp_bdds* b);
uint64_t bddilp_bdds_nr_variables(const bddilp_bdds* b);
const bddmma_instruction* bddilp_bdds_instructions(const bddilp_bdds* b); /* [nr_instructions] */
const uint64_t* bddilp_bdds_delimiters(const bddilp_bdds* b);             /* [nr_bdds + 1] */

/* Text exports of a BDD collection given as flat arrays (the driver's "export bdd lp" / "export bdd graph",
   reference src/bdd_solver/bdd_solver.cpp:400-410 and :432-462 -> include/bdd_collection/bdd_collection.h:731-830 and :663-729).
   write_bdd_lp: the network-flow LP over the arcs of all BDDs, linked by the original variables, with objective costs[0 .. n_costs).
   export_graphviz: BDD bdd_nr as a Graphviz digraph with one cluster per variable.  Both write `path`; 0 on success.
   The caller guarantees delims[0 .. n_bdds] and instr[0 .. delims[n_bdds]) (no array length crosses the ABI); the arrays are
   validated (ascending delimiters, terminals last, forward arcs inside each BDD) BEFORE `path` is opened, so a malformed
   collection leaves an existing file untouched. */
int bddilp_write_bdd_lp(const bddmma_instruction* instr, const uint64_t* delims, uint64_t n_bdds, const double* costs, uint64_t n_costs,
                        const char* path);
int bddilp_export_graphviz(const bddmma_instruction* instr, const uint64_t* delims, uint64_t n_bdds, uint64_t bdd_nr, const char* path);


/* The benchmark's synthetic instance: random set cover, n_rows rows of k distinct variables, costs U(1,10)
 * (0 for variables in no row), all drawn from one std::mt19937_64(seed) — draw order in bdd_amd/csrc/host/instances.cpp.
 * rows: uint64[n_rows * k] (each row sorted), costs: double[n_vars]. */
int bddilp_random_set_cover(uint64_t n_vars, uint64_t n_rows, uint64_t k, uint64_t seed, uint64_t* rows, double* costs);

#ifdef __cplusplus
}
#endif
#endif

// solver_f64.hip — SolverT<double> (solver_impl.hpp) and every kernel it launches, as one translation unit.
#include "solver_impl.hpp"

namespace bddmma {
SolverBase* make_solver_f64() { return new SolverT<double>(); }
}  // namespace bddmma

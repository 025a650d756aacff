// solver_f32.hip — SolverT<float> (solver_impl.hpp) and every kernel it launches, as one translation unit.
#include "solver_impl.hpp"

namespace bddmma {
SolverBase* make_solver_f32() { return new SolverT<float>(); }
}  // namespace bddmma

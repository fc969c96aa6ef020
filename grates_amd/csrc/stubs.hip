// Entry points that are declared in include/shg.h but not implemented yet: they fail loudly.
#include "common.h"
using namespace shg;

namespace shg {
void plan_free_aux(shg_plan*) {}
}

#define SHG_TODO(name) return shg::fail(SHG_ERR_UNSUPPORTED, name ": not implemented in this build")

extern "C" int shg_synthesis_points(int, const double*, const double*, const double*, int, const double*, int, double*, void*) { SHG_TODO("shg_synthesis_points"); }
extern "C" int shg_covprop_diag(shg_plan*, const double*, int, int, int, double*, void*) { SHG_TODO("shg_covprop_diag"); }
extern "C" int shg_covprop_points(int, const double*, const double*, const double*, int, const double*, int, double*, void*) { SHG_TODO("shg_covprop_points"); }
extern "C" int shg_orderwise_filter(const double*, const int64_t*, int, int, const double*, int, double*, void*) { SHG_TODO("shg_orderwise_filter"); }
extern "C" int shg_dense_filter(const double*, int, const double*, int, double*, void*) { SHG_TODO("shg_dense_filter"); }
extern "C" int shg_dgemm(int, int, int, const double*, int, const double*, int, double*, int, void*) { SHG_TODO("shg_dgemm"); }
extern "C" int shg_analysis(shg_plan*, const double*, const double*, int, int, double*, void*) { SHG_TODO("shg_analysis"); }

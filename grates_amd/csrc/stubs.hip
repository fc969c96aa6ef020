// Entry points that are declared in include/shg.h but not implemented yet: they fail loudly.
#include "common.h"
using namespace shg;

namespace shg {
void plan_free_aux(shg_plan*) {}
}

#define SHG_TODO(name) return shg::fail(SHG_ERR_UNSUPPORTED, name ": not implemented in this build")

extern "C" int shg_analysis(shg_plan*, const double*, const double*, int, int, double*, void*) { SHG_TODO("shg_analysis"); }

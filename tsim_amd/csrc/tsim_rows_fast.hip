// tsim_rows_fast.hip - row kernels of the exact-value ("fast") formulation.
#define TSIM_ROWS_FAST true
#define TSIM_ROWS_NAME(sym) sym##_fast
#include "tsim_rows_impl.hip.h"

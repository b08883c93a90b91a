// tsim_rows_faithful.hip - row kernels of the faithful formulation (int32 mirror of the reference).
#define TSIM_ROWS_FAST false
#define TSIM_ROWS_NAME(sym) sym##_faithful
#include "tsim_rows_impl.hip.h"

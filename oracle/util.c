/* oracle/util.c — thread-count control for the timed cpu_baseline leg. TEST INFRASTRUCTURE ONLY. */
#include <omp.h>
int orc_max_threads(void) { return omp_get_max_threads(); }
void orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

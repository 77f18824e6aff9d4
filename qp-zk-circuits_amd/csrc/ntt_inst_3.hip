// ntt_inst_3.hip — instantiations of the NTT pass kernel (see ntt_kernel_impl.hpp)
#include "ntt_kernel_impl.hpp"
NTT_DEFINE_CASE(5, 5)

// ntt_inst_2.hip — instantiations of the NTT pass kernel (see ntt_kernel_impl.hpp)
#include "ntt_kernel_impl.hpp"
NTT_DEFINE_CASE(4, 4)
NTT_DEFINE_CASE(5, 4)

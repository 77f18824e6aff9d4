// ntt_inst_1.hip — instantiations of the NTT pass kernel (see ntt_kernel_impl.hpp)
#include "ntt_kernel_impl.hpp"
NTT_DEFINE_CASE(3, 2)
NTT_DEFINE_CASE(3, 3)
NTT_DEFINE_CASE(4, 3)

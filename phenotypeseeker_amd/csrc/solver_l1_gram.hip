// FORM 0 of the bit-packed L1-logistic kernel: see solver_l1_bits.h
#include "solver_l1_bits.h"

hipError_t psk_l1_bits_launch_gram(const psk_l1_bits_launch &a) { return l1_bits_launch<0>(a); }

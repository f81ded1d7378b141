// fused ELBO kernel instantiations for template ability width 8
#define VIBO_AT 8
#include "vibo_elbo_inst.inc"

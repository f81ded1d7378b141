// fused ELBO kernel instantiations for template ability width 1
#define VIBO_AT 1
#include "vibo_elbo_inst.inc"

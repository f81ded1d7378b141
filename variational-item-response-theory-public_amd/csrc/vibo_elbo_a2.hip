// fused ELBO kernel instantiations for template ability width 2
#define VIBO_AT 2
#include "vibo_elbo_inst.inc"

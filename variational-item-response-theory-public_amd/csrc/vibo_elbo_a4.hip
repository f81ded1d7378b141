// fused ELBO kernel instantiations for template ability width 4
#define VIBO_AT 4
#include "vibo_elbo_inst.inc"

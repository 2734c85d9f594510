// Stages not implemented yet fail loudly (no CPU fallback).
#include "common.h"
int cdm_synth_impl(cdm_ctx *, uint64_t, uint64_t, uint64_t, uint32_t, uint32_t, uint64_t, cdm_seqdb **) { cdm_set_error("cdm_seqdb_synth: not implemented yet"); return CDM_ERR_UNSUPPORTED; }

// ancient_contig_merge's queue and extension loop on the device (contigqueue.hip), as contig.hip calls it.
#pragma once
#include <cstdint>
#include <vector>

#include "common.h"

struct CqResult {
    cdm_seqdb *grown = nullptr;             // the contigs that grew, in the order of their queries (NULL: none); keys and lengths set, wasExtended = 1
    std::vector<uint32_t> grownIdx;         // their queries (ascending)
    std::vector<uint8_t> outExt;            // wasExtended of every sequence of the result (a query handed back: its input flag, to be replaced)
    std::vector<uint8_t> handedBack;        // empty, or a byte per query: 1 = the host decides this query (contigqueue.hip, top)
    uint32_t nHandedBack = 0;
};
// meta / owner / dStats: cdm_build_meta's table, the query of every record, k_contig_stats' counts (device memory)
int cdm_contig_queue_device(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_alns *alns, const cdm_ancient_params *par, float mergeSeqIdThr, const SeqMeta *meta, const uint32_t *owner,
                            const ContigStat *dStats, CqResult *res);
// have the tables of the C library's lgammaf / logf been filled and copied to this device already (about a second, once per process)?
bool cdm_contig_tables_ready(int device);

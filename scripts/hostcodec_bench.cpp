// Probe (not product): the host text codecs of the module binary timed without a device - DB load, prefilter / alignment text
// parse and format, DB write - on DB files made by any implementation of the modules.
//   g++ -O2 -std=c++17 -fopenmp -Iinclude scripts/hostcodec_bench.cpp carpedeam_amd/csrc/host/{mmdb,ingest}.cpp -Lcarpedeam_amd -lcarpedeam_hip -lz -Wl,-rpath,$PWD/carpedeam_amd -o /tmp/hostcodec
//   /tmp/hostcodec <seqDB> <prefDB> <alnDB> <outdir> [threads]
#define main module_main
#include "../carpedeam_amd/csrc/host/main.cpp"
#undef main
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const int th = argc > 5 ? atoi(argv[5]) : 8;
    omp_set_dynamic(0); omp_set_num_threads(th);
    std::string err, out = argv[4];
    double t = now();
    auto lap = [&](const char *w) { const double n = now(); printf("  %-36s %.3f s\n", w, n - t); t = n; };
    MmDb seq, pref, aln;
    if (!seq.load(argv[1], &err)) { puts(err.c_str()); return 1; } lap("seq DB load");
    if (!pref.load(argv[2], &err)) { puts(err.c_str()); return 1; } lap("pref DB load");
    if (!aln.load(argv[3], &err)) { puts(err.c_str()); return 1; } lap("aln DB load");
    HVec<uint64_t> off; HVec<cdm_hit> rec;
    parsePrefDb(pref, seq, off, rec); lap("pref parse");
    printf("    %zu hits\n", rec.size());
    { std::vector<OutChunk> chunks; formatPrefDb(seq, off.data(), rec.data(), chunks); lap("pref format");
      if (!mmdbWriteChunks(out + "/pref", 14, chunks, &err)) return 1; lap("pref write"); }
    HVec<uint64_t> aoff; HVec<cdm_aln> arec;
    parseAlnDb(aln, seq, aoff, arec); lap("aln parse");
    printf("    %zu records\n", arec.size());
    uint64_t dbRes = 0; for (size_t i = 0; i < seq.size(); i++) dbRes += seq.len[i] - 2;
    for (auto &r : arec) { const int alnLen = std::max(abs(r.q_end - r.q_start), abs(r.db_end - r.db_start)) + 1; r.ident = (int) (r.seq_id * alnLen + 0.5f); }
    t = now();
    { std::vector<OutChunk> chunks; formatAlnDb(seq, pref, aoff.data(), arec.data(), dbRes, chunks); lap("aln format");
      if (!mmdbWriteChunks(out + "/aln", 5, chunks, &err)) return 1; lap("aln write"); }
    return 0;
}

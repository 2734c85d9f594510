// Host side of cdm_damage_load: the 11 + 11 damage matrices and the log look-up tables the kernels read.
// Compiled with g++ (x87 long double) like the reference, because the tables' last bits decide arg-max ties.
//
// What is reproduced (reference file:line):
//   profile parsing          src/assembler/nuclassembleUtil.h:53-102 (12 tab separated columns, header line,
//                            values read with operator>> into long double)
//   matrix construction      src/assembler/nuclassembleUtil.cpp:821-1007 incl. its quirks: off-diagonals summed in
//                            double, diagonal = 1.0 - sum (double); the interior "default" matrix takes C>T from the
//                            LAST 5' row and G>A from the FIRST 3' row; those four defaults pass through double;
//                            3' rows fill slots 6..10 in file order; reverse matrices swap C>T/C>C with G>A/G>G of
//                            the mirrored slot.
//   sequencing error matrix  src/assembler/nuclassembleUtil.cpp:49-65
//   log terms                src/assembler/correction.cpp:48-77,93-107 and nuclassembleUtil.cpp:259-276
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../common.h"

typedef long double ld;
namespace {
struct Rates { ld s[12]; };
struct Mat { ld p[4][4]; };
const double SMOOTHING = 0.001;

bool readProfile(const std::string &fn, std::vector<Rates> &rows, std::string *err) {
    std::ifstream f(fn.c_str());
    if (!f.good()) { *err = "cannot open damage profile " + fn; return false; }
    auto split = [](const std::string &l) {
        std::vector<std::string> t; std::string cur;
        for (size_t i = 0; i < l.size(); i++) { if (l[i] == '\t') { t.push_back(cur); cur.clear(); } else cur += l[i]; }
        t.push_back(cur); return t;
    };
    std::string line;
    if (!std::getline(f, line) || split(line).size() != 12) { *err = "Profile not 12 fields: " + fn; return false; }
    while (std::getline(f, line)) {
        std::vector<std::string> t = split(line);
        if (t.size() != 12) { *err = "Profile not 12 fields: " + fn; return false; }
        Rates r;
        for (int k = 0; k < 12; k++) { std::istringstream in(t[k]); ld v = 0; in >> v; r.s[k] = v; }
        rows.push_back(r);
    }
    return true;
}

void seqErrMatrix(Mat &m, ld err) {
    for (int o = 0; o < 4; o++) for (int b = 0; b < 4; b++) m.p[o][b] = (o == b) ? 1 - err : err / 3;
}
}  // namespace

int cdm_build_damage(const char *prefixC, long double mats[2][11][4][4], DamageLut *lut, std::string *err) {
    const std::string prefix = prefixC ? prefixC : "";
    std::vector<Rates> sub5, sub3;
    if (prefix.empty()) {  // "3p.prof"/"5p.prof" without a prefix: no damage (nuclassembleUtil.cpp:824-832)
        Rates z; for (int k = 0; k < 12; k++) z.s[k] = 0.0;
        sub5.assign(5, z); sub3.assign(5, z);
    } else {
        if (!readProfile(prefix + "5p.prof", sub5, err) || !readProfile(prefix + "3p.prof", sub3, err)) return CDM_ERR_IO;
    }
    if (sub5.size() + sub3.size() < 5 || sub5.empty() || sub3.empty()) { *err = "damage profiles need at least 5 rows in total"; return CDM_ERR_IO; }
    Mat def;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) def.p[i][j] = (i == j) ? 1.0f : 0.0f;
    def.p[1][3] = sub5.back().s[5];  def.p[1][1] = 1 - sub5.back().s[5];
    def.p[2][0] = sub3.front().s[6]; def.p[2][2] = 1 - sub3.front().s[6];
    const double dCC = def.p[1][1], dCT = def.p[1][3], dGA = def.p[2][0], dGG = def.p[2][2];
    std::vector<Mat> sub;
    for (size_t r = 0; r < sub5.size() + sub3.size(); r++) {
        const bool five = r < sub5.size();
        const Rates &o = five ? sub5[r] : sub3[r - sub5.size()];
        Mat m; int idx = 0;
        for (int i = 0; i < 4; i++) {
            double sum = 0.0;
            for (int j = 0; j < 4; j++) { if (i == j) continue; m.p[i][j] = o.s[idx]; sum += o.s[idx]; idx++; }
            m.p[i][i] = 1.0 - sum;
        }
        if (five) { m.p[2][0] = dGA; m.p[2][2] = dGG; } else { m.p[1][1] = dCC; m.p[1][3] = dCT; }
        sub.push_back(m);
    }
    Mat all[11], rev[11];
    for (int i = 0; i < 11; i++) all[i] = def;
    for (int i = 0; i < 5; i++) all[i] = sub[i];
    for (int i = 0; i < 5; i++) all[6 + i] = sub[sub.size() - 5 + i];
    for (int i = 0; i < 11; i++) {
        rev[i] = all[i];
        const Mat &e = all[10 - i];
        rev[i].p[1][3] = e.p[2][0]; rev[i].p[1][1] = e.p[2][2];
        rev[i].p[2][0] = e.p[1][3]; rev[i].p[2][2] = e.p[1][1];
    }
    for (int i = 0; i < 11; i++) for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) { mats[0][i][a][b] = all[i].p[a][b]; mats[1][i][a][b] = rev[i].p[a][b]; }

    // ---- ancient_correction tables, seqErr = 0.01 (correction.cpp:196)
    Mat e01; seqErrMatrix(e01, 0.01L);
    for (int qb = 0; qb < 4; qb++)
        for (int t = 0; t < 4; t++) lut->logT[qb][t] = std::log(e01.p[t][qb]);  // logl, rounded to double on store
    for (int c = 0; c < 11; c++)
        for (int qb = 0; qb < 4; qb++)
            for (int q = 0; q < 4; q++) {
                double deam = all[c].p[q][qb];
                lut->logQ[c][qb][q] = std::log(std::max(deam, SMOOTHING));
            }
    for (int qb = 0; qb < 4; qb++) for (int q = 0; q < 4; q++) lut->logQ[11][qb][q] = std::log(e01.p[q][qb]);
    for (int r = 0; r < 2; r++)
        for (int l = 0; l < 11; l++)
            for (int q = 0; q < 4; q++)
                for (int t = 0; t < 4; t++) {
                    double d = r ? rev[l].p[q][t] : all[l].p[q][t];
                    lut->logD[r][l][q][t] = std::log(std::max(d, SMOOTHING));
                }
    // ---- ancient_read_assemble per-column likelihood, seqErr = 0.001 (ancientReadsResults.cpp:172)
    Mat e001; seqErrMatrix(e001, 0.001L);
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 11; c++)
            for (int qb = 0; qb < 4; qb++)
                for (int tb = 0; tb < 4; tb++) {
                    const Mat &tp = r ? rev[c] : all[c];
                    double lik = 0;
                    for (int x = 0; x < 4; x++) {
                        double match = std::max(static_cast<ld>(SMOOTHING), tp.p[qb][x]);
                        ld tErr = e001.p[x][tb];
                        lik += (tErr * match);
                    }
                    lut->logLik[r][c][qb][tb] = log(lik);
                }
    return CDM_OK;
}

// Host side E-value / bit score of rescorediagonal (R3 in SURVEY.md 8(a)).
//   lib/mmseqs/src/alignment/EvalueComputation.h:18-24,36-40 over ALP:
//   lib/mmseqs/lib/alp/sls_alignment_evaluer.cpp:989-1025 (area), sls_alignment_evaluer.hpp:154-162 (evaluePerArea,
//   bitScore), sls_pvalues.cpp:366-541 (finite size correction), sls_basic.hpp:195-198 (normal_probability).
// The gapless Gumbel parameters are what ALP's initGapless derives for lib/mmseqs/data/nucleotide.out with its
// background frequencies; they are constants of that matrix (hex-exact from the compiled reference, fixture
// tests/golden/functions/alp.txt).  Other scoring matrices are not supported on this path.
// Compiled with g++ -O3 -march=x86-64-v3 like the reference recipe: the build contracts a*b+c into FMAs.
#include <algorithm>
#include <climits>
#include <cmath>
#include <vector>

#include "../common.h"

namespace {
const double LAMBDA = 0x1.4478764a1b24ap-1, K = 0x1.a1c1e68ea2ab1p-2, A_IJ = 0x1.639ba57df0ecdp-1, ALPHA_IJ = 0x1.aaaae7ad40e75p-1,
             SIGMA = 0x1.aaaae7ad40e75p-1;
inline double normalProb(double x) { return 0.5 * erfc(-sqrt(0.5) * x); }

double area(double y, double m, double n) {
    const double pi = 3.1415926535897932384626433832795;
    const double constVal = 1 / sqrt(2.0 * pi);
    const double viThr = std::max(2.0 * ALPHA_IJ / LAMBDA, 0.0), cThr = std::max(2.0 * SIGMA / LAMBDA, 0.0);
    double tmp = A_IJ * y + 0.0;
    double mLi = m - tmp;
    double viY = std::max(viThr, ALPHA_IJ * y + 0.0);
    double sVi = sqrt(viY);
    double mF = (sVi == 0.0) ? 1e100 : mLi / sVi;
    double PmF = normalProb(mF);
    double EmF = -constVal * exp(-0.5 * mF * mF);
    double mLiP = mLi * PmF;
    double sViE = sVi * EmF;
    double p1 = mLiP - sViE;
    tmp = A_IJ * y + 0.0;
    double nLj = n - tmp;
    double vjY = std::max(viThr, ALPHA_IJ * y + 0.0);
    double sVj = sqrt(vjY);
    double nF = (sVj == 0.0) ? 1e100 : nLj / sVj;
    double PnF = normalProb(nF);
    double EnF = -constVal * exp(-0.5 * nF * nF);
    double nLjP = nLj * PnF;
    double sVjE = sVj * EnF;
    double p2 = nLjP - sVjE;
    double cY = std::max(cThr, SIGMA * y + 0.0);
    double PP = PmF * PnF;
    double cPP = cY * PP;
    double p1p2 = p1 * p2;
    return p1p2 + cPP;
}
const double LOGK = log(K);
}  // namespace

double cdm_evalue_host(double rawScore, double qLen, uint64_t dbResidues) {
    const double epa = K * exp(-LAMBDA * rawScore);
    return epa * area(rawScore, (double) dbResidues, qLen);
}
int cdm_bit_score_host(double rawScore) { return static_cast<int>(std::fma(LAMBDA, rawScore, -LOGK) / log(2.0) + 0.5); }

void cdm_min_score_table(double evalThr, uint64_t dbResidues, uint32_t maxLen, std::vector<int32_t> &table) {
    table.assign((size_t) maxLen + 1, INT_MAX);
    for (uint32_t L = 1; L <= maxLen; L++) {
        // E-value is decreasing in the score over [0, 2L]; find the first score that passes
        int lo = 0, hi = 2 * (int) L;
        if (!(cdm_evalue_host(hi, L, dbResidues) <= evalThr)) continue;
        while (lo < hi) {
            int mid = (lo + hi) / 2;
            if (cdm_evalue_host(mid, L, dbResidues) <= evalThr) hi = mid; else lo = mid + 1;
        }
        table[L] = lo;
    }
}

// ---- gapped alignments (`align`, lib/mmseqs/src/alignment/Alignment.cpp:273: EvalueComputation(dbSize, m, gapOpen, gapExtend)).  For
// nucleotide.out with --gap-open 5 --gap-extend 2 (what `ancient_assemble` hands its linclust, GuidedNuclassembler.cpp) the reference
// has no tabulated parameters (EvalueComputation.h:61-82 lists 7 / 1 only) and lets ALP estimate them by its importance-sampling
// simulation (initGapped, tolerances 0.01 / 0.05, seed 42): the twelve Gumbel parameters and the three thresholds below are that
// simulation's result, taken hex-exact from the reference's object code (oracle/ref_driver.cpp `probe alp <matrix> 5 2`, fixture
// tests/golden/functions/alp_gapped_5_2.txt) - the simulation itself is not restated.  area() is ALP's finite-size formula with the
// intercepts b, beta, tau that vanish in the gapless case (sls_pvalues.cpp:366-541, statement order kept: FMA contraction).
namespace {
struct Gumbel { double lambda, K, aI, aJ, alphaI, alphaJ, sigma, bI, bJ, betaI, betaJ, tau, viThr, vjThr, cThr; };
const Gumbel GAPPED_5_2 = {0x1.3de995e742594p-1, 0x1.6837f664724adp-2, 0x1.7d956af2b60d7p-1, 0x1.7d956af2b60d7p-1, 0x1.037654c94bd2fp+0, 0x1.037654c94bd2fp+0,
                           0x1.00d16215dfda7p+0, -0x1.6ba8cc62c7c8cp-1, -0x1.6ba8cc62c7c8cp-1, -0x1.42e626a2af9bp+1, -0x1.42e626a2af9bp+1, -0x1.306383babbcf8p+1,
                           0x1.a1dd95922d781p+1, 0x1.a1dd95922d781p+1, 0x1.9d9b5aeaf525cp+1};
double areaOf(const Gumbel &g, double y, double m, double n) {
    const double pi = 3.1415926535897932384626433832795;
    const double constVal = 1 / sqrt(2.0 * pi);
    double tmp = g.aI * y + g.bI;
    double mLi = m - tmp;
    double viY = std::max(g.viThr, g.alphaI * y + g.betaI);
    double sVi = sqrt(viY);
    double mF = (sVi == 0.0) ? 1e100 : mLi / sVi;
    double PmF = normalProb(mF);
    double EmF = -constVal * exp(-0.5 * mF * mF);
    double mLiP = mLi * PmF;
    double sViE = sVi * EmF;
    double p1 = mLiP - sViE;
    tmp = g.aJ * y + g.bJ;
    double nLj = n - tmp;
    double vjY = std::max(g.vjThr, g.alphaJ * y + g.betaJ);
    double sVj = sqrt(vjY);
    double nF = (sVj == 0.0) ? 1e100 : nLj / sVj;
    double PnF = normalProb(nF);
    double EnF = -constVal * exp(-0.5 * nF * nF);
    double nLjP = nLj * PnF;
    double sVjE = sVj * EnF;
    double p2 = nLjP - sVjE;
    double cY = std::max(g.cThr, g.sigma * y + g.tau);
    double PP = PmF * PnF;
    double cPP = cY * PP;
    double p1p2 = p1 * p2;
    return p1p2 + cPP;
}
const double LOGK_GAPPED_5_2 = log(GAPPED_5_2.K);
}  // namespace
bool cdm_gapped_costs_known(int gapOpen, int gapExtend) { return gapOpen == 5 && gapExtend == 2; }
double cdm_evalue_gapped_host(double rawScore, double qLen, uint64_t dbResidues) {
    const Gumbel &g = GAPPED_5_2;
    const double epa = g.K * exp(-g.lambda * rawScore);
    return epa * areaOf(g, rawScore, (double) dbResidues, qLen);
}
int cdm_bit_score_gapped_host(double rawScore) { return static_cast<int>(std::fma(GAPPED_5_2.lambda, rawScore, -LOGK_GAPPED_5_2) / log(2.0) + 0.5); }

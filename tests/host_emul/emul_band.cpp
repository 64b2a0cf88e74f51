// Host emulation of the band kernel: runs the barrier-separated phases of dsp_amd/csrc/band_phases.h for
// tid = 0..nt-1 in a loop.  TEST INFRASTRUCTURE ONLY -- it exists so the kernels' index arithmetic can be
// checked against the oracle on a machine without a GPU; the product never loads it.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../dsp_amd/csrc/tables.h"

using namespace echo;

extern "C" int emul_band(int N, int M, int K, int nb, int min_delay, int noise_n, int detector, double gate,
                         const double* taps, const double* chirps, int mode, const float* rx, const double* d_in,
                         double* corr_io, int B, double* band_out, float* spec_out, int spec_db, int do_corr,
                         double* snr, double* amp, int* peak) {
    BandParams p;
    std::memset(&p, 0, sizeof(p));
    if (!plan_band(p, N, M, K, nb, min_delay, noise_n, detector, gate)) return -2;
    HostTables t;
    build_tables(t, p, taps, chirps);
    p.taps_rev = t.taps_rev.data(); p.chirp = t.chirp.data();
    p.bw = t.bw.data(); p.bhat = t.bhat.data(); p.tw = t.tw.data(); p.unpack = t.unpack.data();
    p.mode = mode; p.do_corr = do_corr; p.do_spec = spec_out != nullptr; p.spec_db = spec_db;
    p.rx_in = rx; p.d_in = d_in; p.corr_io = corr_io; p.band_out = band_out; p.spec_out = spec_out;
    p.write_corr_mutated = 1;
    std::vector<BandRes> res((size_t)B * nb);
    p.res_out = res.data();
    const int nt = kNT;
    std::vector<unsigned char> smem(p.smem_bytes);
    for (int blk = 0; blk < B * nb; ++blk) {
        std::memset(smem.data(), 0xff, smem.size());   // poison: NaN patterns expose reads of unwritten LDS
        unsigned char* s = smem.data();
#define PHASE(call) for (int tid = 0; tid < nt; ++tid) { call; }
        PHASE(ph_load(tid, nt, blk, p, s));
        PHASE(ph_fir(tid, nt, blk, p, s));
        PHASE(ph_corr(tid, nt, blk, p, s));
        PHASE(ph_reduce_a(tid, nt, p, s));
        PHASE(ph_reduce_b(tid, nt, p, s));
        PHASE(ph_rank(tid, nt, p, s));
        PHASE(ph_finish(tid, nt, blk, p, s));
        PHASE(ph_spec_load(tid, nt, p, s));
        PHASE(ph_fft_r8<false>(tid, nt, 1024, p, s));
        PHASE(ph_fft_r8<false>(tid, nt, 128, p, s));
        PHASE(ph_fft_r8<false>(tid, nt, 16, p, s));
        PHASE(ph_fft_mid(tid, nt, p, s));
        PHASE(ph_fft_r8<true>(tid, nt, 16, p, s));
        PHASE(ph_fft_r8<true>(tid, nt, 128, p, s));
        PHASE(ph_fft_r8<true>(tid, nt, 1024, p, s));
        PHASE(ph_spec_store(tid, nt, blk, p, s));
#undef PHASE
    }
    for (size_t i = 0; i < res.size(); ++i) {
        if (snr) snr[i] = res[i].snr_db;
        if (amp) amp[i] = res[i].amplitude;
        if (peak) peak[i] = res[i].peak_idx;
    }
    return 0;
}

// Host emulation of the band kernel: runs the barrier-separated phases of dsp_amd/csrc/band_phases.h for
// tid = 0..nt-1 in a loop.  TEST INFRASTRUCTURE ONLY -- it exists so the kernels' index arithmetic can be
// checked against the oracle on a machine without a GPU; the product never loads it.
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "../../dsp_amd/csrc/tables.h"

using namespace echo;

extern "C" int emul_band(int N, int M, int K, int nb, int min_delay, int noise_n, int detector, double gate,
                         const double* taps, const double* chirps, int mode, const float* rx, const double* d_in,
                         double* corr_io, int B, double* band_out, float* spec_out, int spec_db, int do_corr,
                         double* snr, double* amp, int* peak, int corr_fast) {
    BandParams p;
    std::memset(&p, 0, sizeof(p));
    if (!plan_band(p, N, M, K, nb, min_delay, noise_n, detector, gate)) return -2;
    HostTables t;
    build_tables(t, p, taps, chirps);
    p.taps_rev = t.taps_rev.data(); p.chirp = t.chirp.data();
    p.bw = t.bw.data(); p.bhat = t.bhat.data(); p.tw = t.tw.data(); p.post = t.post.data();
    p.mode = mode; p.do_corr = do_corr; p.spec_db = spec_db;
    std::vector<float> bandf(spec_out ? (size_t)B * nb * N : 0);
    if (spec_out && (mode == MODE_RX_FIR || mode == MODE_Y_FIR)) { p.bandf_out = bandf.data(); p.specf_in = bandf.data(); }
    else if (spec_out) p.specd_in = d_in;
    p.rx_in = rx; p.d_in = d_in; p.corr_io = corr_io; p.band_out = band_out; p.spec_out = spec_out;
    p.write_corr_mutated = 1;
    p.corr_fast = corr_fast;
    if (corr_fast) { p.corr_io = nullptr; p.band_out = nullptr; }
    std::vector<BandRes> res((size_t)B * nb);
    p.res_out = res.data();
    const int nt = kNT;
    std::vector<unsigned char> smem(std::max(p.smem_bytes_host, std::max(p.spec_smem_bytes, kFftM * 8)));
    struct FirAcc { double v[kR]; };
    std::vector<FirAcc> fir(nt);
    for (int blk = 0; blk < B * nb; ++blk) {
        std::memset(smem.data(), 0xff, smem.size());   // poison: NaN patterns expose reads of unwritten LDS
        unsigned char* s = smem.data();
#define PHASE(call) for (int tid = 0; tid < nt; ++tid) { call; }
        PHASE(ph_load(tid, nt, blk, p, s));
        PHASE(ph_fir(tid, nt, blk, p, s, fir[tid].v));
        PHASE(ph_fir_store(tid, nt, blk, p, s, fir[tid].v));
        PHASE(ph_corr(tid, nt, blk, p, s));
        PHASE(ph_reduce(tid, nt, p, s));
        PHASE(ph_rank_count(tid, nt, p, s));
        PHASE(ph_rank_pick(tid, nt, p, s));
        PHASE(ph_finish(tid, nt, blk, p, s));
        if (spec_out) {
            std::memset(smem.data(), 0xff, smem.size());
            // per-thread register state that the kernel carries across barriers
            std::vector<Tw3> tw1024(nt), tw128(nt), tw16(nt);
            std::vector<Bh8> bh(nt);
            PHASE(tw1024[tid] = fft_tw_load(tid, 1024, p); tw128[tid] = fft_tw_load(tid, 128, p);
                  tw16[tid] = fft_tw_load(tid, 16, p); bh[tid] = fft_bh_load(tid, p));
            PHASE(ph_spec_load(tid, nt, blk, p, s));
            PHASE(ph_fft_r8<false>(tid, 1024, p, s, tw1024[tid]));
            PHASE(ph_fft_r8<false>(tid, 128, p, s, tw128[tid]));
            PHASE(ph_fft_r8<false>(tid, 16, p, s, tw16[tid]));
            PHASE(ph_fft_mid(tid, p, s, bh[tid]));
            PHASE(ph_fft_r8<true>(tid, 16, p, s, tw16[tid]));
            PHASE(ph_fft_r8<true>(tid, 128, p, s, tw128[tid]));
            PHASE(ph_fft_r8<true>(tid, 1024, p, s, tw1024[tid]));
            PHASE(ph_spec_store(tid, nt, blk, p, s));
        }
#undef PHASE
    }
    for (size_t i = 0; i < res.size(); ++i) {
        if (snr) snr[i] = res[i].snr_db;
        if (amp) amp[i] = res[i].amplitude;
        if (peak) peak[i] = res[i].peak_idx;
    }
    return 0;
}

// Long-frame path (dsp_amd/csrc/long_phases.h): FIR tiles -> matched-filter tiles -> median/MAD detector.
// y: [B][nb][N] zero-phase IIR output.  band_out [B][nb][N]; corr_out [B][nb][L] (lags below k_begin are left
// untouched); peak [B][nb].  With corr_in non-null the FIR and correlation are skipped and the detector runs on
// corr_in in place (blind zone written back), like a direct first_strong_peak call.
#include "../../dsp_amd/csrc/long_phases.h"

extern "C" int emul_long(int N, int M, int K, int nb, int min_delay, int detector, const double* taps,
                         const double* chirps, const double* y, double* corr_in, int B, int k_begin,
                         double* band_out, double* corr_out, int* peak) {
    LongParams p;
    std::memset(&p, 0, sizeof(p));
    if (!plan_long(p, N, M, K, nb, min_delay, detector)) return -2;
    HostTables t;
    build_coef_tables(t, nb, K, M, taps, chirps);
    p.taps_rev = t.taps_rev.data(); p.chirp = t.chirp.data();
    std::vector<BandRes> res((size_t)B * nb);
    p.res_out = res.data();
    const int smem_bytes = std::max(p.fir_smem, std::max(p.corr_smem, p.det_smem));
    std::vector<unsigned char> smem(smem_bytes);
    unsigned char* s = smem.data();
#define PHASE(...) for (int tid = 0; tid < nt; ++tid) { __VA_ARGS__; }
#define SYNC
    for (int blk = 0; blk < B * nb; ++blk) {
        if (!corr_in) {
            p.fir_rx = 0; p.d_in = y; p.band_out = band_out;
            int nt = kLFirNT;
            const int ftiles = (N + nt * kR - 1) / (nt * kR);
            for (int tile = 0; tile < ftiles; ++tile) {
                std::memset(s, 0xff, smem.size());
                std::vector<double> accs((size_t)nt * kR);
                PHASE(ph_lfir_load(tid, nt, tile, blk, p, s));
                PHASE(ph_lfir_run(tid, nt, tile, blk, p, s, *reinterpret_cast<double (*)[kR]>(&accs[(size_t)tid * kR])));
                PHASE(ph_lfir_stage(tid, nt, s, *reinterpret_cast<const double (*)[kR]>(&accs[(size_t)tid * kR])));
                PHASE(ph_lfir_store(tid, nt, tile, blk, p, s));
            }
            p.band_in = band_out; p.corr_out = corr_out; p.k_begin = k_begin; p.corr_k0 = 0; p.corr_stride = p.L;
            nt = kLCorrNT;
            const int ctiles = (p.L - k_begin + nt * kR - 1) / (nt * kR);
            for (int tile = 0; tile < ctiles; ++tile) {
                std::memset(s, 0xff, smem.size());
                PHASE(ph_lcorr_load(tid, nt, blk, p, s));
                PHASE(ph_lcorr_run(tid, nt, tile, blk, p, s));
            }
        }
        p.det_io = corr_in ? corr_in : corr_out; p.det_stride = p.L; p.det_off = p.half;
        p.write_mutated = corr_in ? 1 : 0;
        const int nt = kLDetNT;
        std::memset(s, 0xff, smem.size());
        ECHO_LDET_SEQUENCE(PHASE, SYNC, p, s, blk, nt);
    }
#undef PHASE
#undef SYNC
    for (size_t i = 0; i < res.size(); ++i) peak[i] = res[i].peak_idx;
    return 0;
}

// Detectors of the long-frame path on given correlations corr[B*nb][L] (whole correlation, mutated like a direct
// first_strong_peak call where the reference mutates): detector 0 median, 1 MAD, 2 mean power, 3 two-peak.
extern "C" int emul_long_detect(int N, int M, int K, int nb, int min_delay, int detector, int noise_n, double gate,
                                double* corr, int B, int* peak, double* snr, double* amp, int* valid) {
    LongParams p;
    std::memset(&p, 0, sizeof(p));
    if (!plan_long(p, N, M, K, nb, min_delay, detector)) return -2;
    const int npos = p.L - p.half;
    int nn = noise_n;
    if (detector == 0 && min_delay + nn > npos) nn = npos > min_delay ? npos - min_delay : 0;
    if (detector == 2 && nn > npos) nn = npos;
    p.noise_n = nn; p.snr_gate = gate;
    std::vector<BandRes> res((size_t)B * nb);
    p.res_out = res.data();
    p.det_io = corr; p.det_stride = p.L; p.det_off = p.half; p.write_mutated = 1;
    std::vector<unsigned char> smem(std::max(p.det_smem, std::max(p.tp_smem, p.lm_smem)));
    unsigned char* s = smem.data();
    const int nt = kLDetNT;
#define PHASE(...) for (int tid = 0; tid < nt; ++tid) { __VA_ARGS__; }
#define SYNC
    for (int blk = 0; blk < B * nb; ++blk) {
        std::memset(s, 0xff, smem.size());
        if (detector == 1) {
            ECHO_LDET_SEQUENCE(PHASE, SYNC, p, s, blk, nt);
        } else if (detector == 3) {
            PHASE(ph_tp_load(tid, nt, blk, p, s));
            PHASE(ph_tp_ref_fold(tid, nt, p, s));
            PHASE(ph_tp_ref(tid, nt, p, s));
            PHASE(ph_tp_second(tid, nt, p, s));
            PHASE(ph_tp_second_fold(tid, nt, p, s));
            PHASE(ph_tp_finish(tid, nt, blk, p, s));
        } else {
            ECHO_LMED_SEQUENCE(PHASE, SYNC, p, s, blk, nt);
        }
    }
#undef PHASE
#undef SYNC
    for (size_t i = 0; i < res.size(); ++i) {
        peak[i] = res[i].peak_idx; snr[i] = res[i].snr_db; amp[i] = res[i].amplitude; valid[i] = res[i].pad;
    }
    return 0;
}

// Matched filter of long frames by FFT (dsp_amd/csrc/cfft_phases.h), thread by thread: corr[k], k_begin <= k < L,
// written to corr_out[blk * L + k].
extern "C" int emul_long_corr_fft(int N, int M, int K, int nb, const double* chirps, const double* band, int B,
                                  int k_begin, int nt, double* corr_out) {
    LongParams p;
    std::memset(&p, 0, sizeof(p));
    if (!plan_long(p, N, M, K, nb, 0, 1)) return -2;
    if (!p.cf_use) return -3;
    std::vector<zd> g, w2, tw;
    cf_tables(nb, M, chirps, g, w2, tw);
    p.cf_g = g.data(); p.cf_w2 = w2.data(); p.cf_tw = tw.data();
    p.band_in = band; p.corr_out = corr_out; p.k_begin = k_begin; p.corr_k0 = 0; p.corr_stride = p.L;
    std::vector<unsigned char> smem(kCfSmem);
    unsigned char* s = smem.data();
#define PHASE(...) for (int tid = 0; tid < nt; ++tid) { __VA_ARGS__; }
#define SYNC
    for (int blk = 0; blk < B * nb; ++blk)
        for (int pair = 0; pair < cf_pairs(p); ++pair) {
            std::memset(s, 0xff, smem.size());
            ECHO_CF_SEQUENCE(PHASE, SYNC, p, s, pair, blk, nt);
        }
#undef PHASE
#undef SYNC
    return cf_pairs(p);
}

// Matched filter of long frames as one real cyclic correlation (dsp_amd/csrc/rcf_phases.h), thread by thread: corr[k],
// k_begin <= k < L, written to corr_out[blk * L + k].  Returns the packed transform length H, or < 0 when the plan
// does not apply.  with_detector != 0: the median/MAD detector (detector 1) or the two-peak detector (3) of the chain runs
// on the transform's LDS layout instead of the store, results into peak / snr / amp.
#include "../../dsp_amd/csrc/rcf_phases.h"
extern "C" int emul_long_rcorr(int N, int M, int K, int nb, const double* chirps, const double* band, int B, int k_begin,
                               int nt, double* corr_out, int detector, int min_delay, int noise_n, double gate,
                               int* peak, double* snr, double* amp, int* valid, int split) {
    LongParams p;
    std::memset(&p, 0, sizeof(p));
    if (!plan_long(p, N, M, K, nb, min_delay, detector ? detector : 1)) return -2;
    if (!plan_rcf(p, k_begin, split != 0)) return -3;
    if (detector && p.rc_nblk != 1) return -4;
    std::vector<zd> wh, wk, roots, g;
    std::vector<int> pos;
    rcf_tables(p, nb, chirps, wh, wk, roots, g, pos);
    p.rc_wh = wh.data(); p.rc_wk = wk.data(); p.rc_roots = roots.data(); p.rc_g = g.data(); p.rc_pos = pos.data();
    p.band_in = band; p.corr_out = corr_out; p.k_begin = k_begin; p.corr_k0 = 0; p.corr_stride = p.L;
    p.noise_n = noise_n; p.snr_gate = gate;
    std::vector<BandRes> res((size_t)B * nb);
    p.res_out = res.data();
    p.det_view = 2; p.det_q0 = k_begin - p.half; p.det_lds_off = p.rc_smem;
    std::vector<unsigned char> smem(p.rc_smem + (kDetExtra > kCfTpExtra ? kDetExtra : kCfTpExtra) + 64);
    unsigned char* s = smem.data();
#define PHASE(...) for (int tid = 0; tid < nt; ++tid) { __VA_ARGS__; }
#define SYNC
    for (int blk = 0; blk < B * nb; ++blk) {
        std::memset(s, 0xff, smem.size());
        if (detector == 1) { ECHO_RC_DET_SEQUENCE(PHASE, SYNC, p, s, blk, nt); }
        else if (detector == 3) { ECHO_RC_TP_SEQUENCE(PHASE, SYNC, p, s, blk, nt); }
        else {
            for (int sub = 0; sub < p.rc_nblk; ++sub) {
                std::memset(s, 0xff, smem.size());
                ECHO_RC_SEQUENCE(PHASE, SYNC, p, s, sub, blk, nt, kGfftMaxBfly);
            }
        }
    }
#undef PHASE
#undef SYNC
    if (detector)
        for (size_t i = 0; i < res.size(); ++i) { peak[i] = res[i].peak_idx; snr[i] = res[i].snr_db; amp[i] = res[i].amplitude; valid[i] = res[i].pad; }
    return split ? p.rc_H * 16 + p.rc_nblk : p.rc_H;
}

// Generic mixed-radix FFT (dsp_amd/csrc/gfft.h), thread by thread.  in/out: n interleaved (re, im) doubles.
#include "../../dsp_amd/csrc/gfft.h"

template <class T> static int run_gfft(int n, int inv, const double* in, double* out, bool merge_small = false) {
    GfftPlan pl;
    if (!gfft_plan(pl, n, 512, merge_small)) return -2;
    std::vector<cpx<T>> roots, wn, a(n), b(n);
    gfft_tables(pl, roots, wn);
    for (int i = 0; i < n; ++i) a[i] = cpx<T>{(T)in[2 * i], (T)in[2 * i + 1]};
    cpx<T>* src = a.data(); cpx<T>* dst = b.data();
    const int nt = 512;
    for (int ps = 0; ps < pl.npass; ++ps) {
        for (int tid = 0; tid < nt; ++tid)
            gfft_pass<T, true>(tid, nt, pl, ps, src, dst, roots.data() + pl.root_off[ps], wn.data(), inv != 0);
        std::swap(src, dst);
    }
    for (int i = 0; i < n; ++i) { out[2 * i] = src[i].x; out[2 * i + 1] = src[i].y; }
    return pl.npass;
}

extern "C" int emul_gfft(int n, int inv, int use_float, const double* in, double* out) {
    return use_float ? run_gfft<float>(n, inv, in, out) : run_gfft<double>(n, inv, in, out);
}
// the same with small factors merged into in-register butterflies (what the spectrum kernels of long frames run)
extern "C" int emul_gfft_bfly(int n, int inv, int use_float, const double* in, double* out) {
    return use_float ? run_gfft<float>(n, inv, in, out, true) : run_gfft<double>(n, inv, in, out, true);
}

// Envelope branch for the correlation (dsp_amd/csrc/env_phases.h), thread by thread.
#include "../../dsp_amd/csrc/env_phases.h"

// allow_lin = 0 keeps the half-length transforms for signals the 4608-point convolution would take
extern "C" int emul_env_opt(int L, int nb, int min_delay, int smooth, const double* corr, int B, double* env_out,
                            int* peak, double* snr, double* amp, double* quality, int* npeaks, int allow_lin) {
    EnvParams p;
    std::memset(&p, 0, sizeof(p));
    if (!plan_env(p, L, nb, min_delay, smooth, allow_lin != 0)) return -2;
    std::vector<cpx<double>> roots, wn, wl, ltw, lwk, lwm, lgm;
    std::vector<double> lgam;
    if (p.lin) {
        envlin_tables(L, ltw, lwk, lgam);
        envlin_mid_tables(lwk, lgam, lwm, lgm);
        p.lin_tw = ltw.data(); p.lin_wk = lwk.data(); p.lin_gam = lgam.data(); p.lin_wmid = lwm.data(); p.lin_gmid = lgm.data();
    } else {
        env_tables(p, roots, wn, wl);
    }
    p.roots = roots.data(); p.wn = wn.data(); p.wl = wl.data();
    std::vector<cpx<double>> bsw, bsb;
    if (p.bs_m) { env_bs_tables(p, bsw, bsb); p.bs_w = bsw.data(); p.bs_bhat = bsb.data(); }
    p.corr = corr; p.env_out = env_out;
    std::vector<EnvRes> res((size_t)B * nb);
    p.res_out = res.data();
    std::vector<unsigned char> smem(p.smem_bytes);
    unsigned char* s = smem.data();
    const int nt = kEnvNT;
    std::vector<EnvPart1> part(nt);
#define PHASE(...) for (int tid = 0; tid < nt; ++tid) { __VA_ARGS__; }
#define SYNC
    for (int blk = 0; blk < B * nb; ++blk) {
        std::memset(s, 0xff, smem.size());
        ECHO_ENV_SEQUENCE(PHASE, SYNC, (void)0, part[tid], p, s, blk, nt);
    }
#undef PHASE
#undef SYNC
    for (size_t i = 0; i < res.size(); ++i) {
        peak[i] = res[i].peak_idx; snr[i] = res[i].snr_db; amp[i] = res[i].amplitude;
        quality[i] = res[i].quality; npeaks[i] = res[i].n_peaks;
    }
    return p.lin ? 2 : 1;
}
extern "C" int emul_env(int L, int nb, int min_delay, int smooth, const double* corr, int B, double* env_out,
                        int* peak, double* snr, double* amp, double* quality, int* npeaks) {
    return emul_env_opt(L, nb, min_delay, smooth, corr, B, env_out, peak, snr, amp, quality, npeaks, 1) > 0 ? 0 : -2;
}

// Fused kernel of the full chain (dsp_amd/csrc/fce_phases.h): FIR of the IIR output y [B][nb][N], correlation as a real
// cyclic convolution (corr_out [B][nb][L], band signal as float32 in bandf_out [B][nb][N]) and the envelope branch on it.
extern "C" int emul_fce(int N, int M, int K, int nb, int min_delay, const double* taps, const double* chirps,
                        const double* y, int B, double* corr_out, float* bandf_out, int* peak, double* snr, double* amp,
                        double* quality, int* npeaks) {
    BandParams bp;
    std::memset(&bp, 0, sizeof(bp));
    if (!plan_band(bp, N, M, K, nb, min_delay, 240, 0, 6.0)) return -2;
    HostTables t;
    build_tables(t, bp, taps, chirps);
    EnvParams p;
    std::memset(&p, 0, sizeof(p));
    if (!plan_env(p, N + M - 1, nb, min_delay, 5) || !plan_fce(p, N, M, K)) return -3;
    std::vector<cpx<double>> ltw, lwk, lwm, lgm, ftw, fwk, fg, fwm, fgm;
    std::vector<double> lgam;
    envlin_tables(p.L, ltw, lwk, lgam);
    envlin_mid_tables(lwk, lgam, lwm, lgm);
    fce_tables(M, nb, chirps, ftw, fwk, fg);
    fce_mid_tables(nb, fwk, fg, fwm, fgm);
    p.lin_tw = ltw.data(); p.lin_wk = lwk.data(); p.lin_gam = lgam.data(); p.lin_wmid = lwm.data(); p.lin_gmid = lgm.data();
    p.fce_tw = ftw.data(); p.fce_wk = fwk.data(); p.fce_g = fg.data(); p.fce_wmid = fwm.data(); p.fce_gmid = fgm.data();
    p.taps_rev = t.taps_rev.data(); p.y_in = y; p.bandf_out = bandf_out; p.corr = corr_out; p.corr_w = corr_out;
    std::vector<EnvRes> res((size_t)B * nb);
    p.res_out = res.data();
    std::vector<unsigned char> smem(p.smem_bytes);
    unsigned char* s = smem.data();
    const int nt = kEnvNT;
    std::vector<EnvPart1> part(nt);
    std::vector<FceKeep> keep(nt);
#define PHASE(...) for (int tid = 0; tid < nt; ++tid) { __VA_ARGS__; }
#define SYNC
    for (int blk = 0; blk < B * nb; ++blk) {
        std::memset(s, 0xff, smem.size());
        ECHO_FCE_SEQUENCE(PHASE, SYNC, (void)0, part[tid], (void)0, keep[tid], p, s, blk, nt);
    }
#undef PHASE
#undef SYNC
    for (size_t i = 0; i < res.size(); ++i) {
        peak[i] = res[i].peak_idx; snr[i] = res[i].snr_db; amp[i] = res[i].amplitude;
        quality[i] = res[i].quality; npeaks[i] = res[i].n_peaks;
    }
    return 0;
}

// Spectrum kernel with the band-signal envelope branch (dsp_amd/csrc/hilb_phases.h): dB spectrum + signal_quality.
#include "../../dsp_amd/csrc/hilb_phases.h"

extern "C" int emul_spec_hilbert(int N, int M, int K, int nb, const double* taps, const double* chirps,
                                 const float* bandf, int B, float* spec_out, double* sq_out, int f9) {
    BandParams p;
    std::memset(&p, 0, sizeof(p));
    if (!plan_band(p, N, M, K, nb, 0, 0, 0, 6.0) || !hilb_fits(p.N, p.n2)) return -2;
    HostTables t;
    build_tables(t, p, taps, chirps);
    p.bw = t.bw.data(); p.bhat = t.bhat.data(); p.tw = t.tw.data(); p.post = t.post.data(); p.uk = t.uk.data();
    p.f9tw9 = t.f9tw9.data(); p.f9twa = t.f9twa.data(); p.f9twb = t.f9twb.data(); p.f9bhat = t.f9bhat.data();
    if (f9 && !p.fft9) return -2;
    p.spec_db = 1; p.specf_in = bandf; p.spec_out = spec_out; p.sq_out = sq_out;
    const int nt = f9 ? kF9NT : kNT;
    std::vector<unsigned char> smem(kFftM * 8 + kHilbRedRows * 8 * nt + 1024);     // one reduction slot per thread on the host
    unsigned char* s = smem.data();
#define PHASE(...) for (int tid = 0; tid < nt; ++tid) { __VA_ARGS__; }
    auto round = [&]() {
        std::vector<Tw3> tw1024(nt), tw128(nt), tw16(nt);
        std::vector<Bh8> bh(nt);
        PHASE(tw1024[tid] = fft_tw_load(tid, 1024, p); tw128[tid] = fft_tw_load(tid, 128, p);
              tw16[tid] = fft_tw_load(tid, 16, p); bh[tid] = fft_bh_load(tid, p));
        PHASE(ph_fft_r8<false>(tid, 1024, p, s, tw1024[tid]));
        PHASE(ph_fft_r8<false>(tid, 128, p, s, tw128[tid]));
        PHASE(ph_fft_r8<false>(tid, 16, p, s, tw16[tid]));
        PHASE(ph_fft_mid(tid, p, s, bh[tid]));
        PHASE(ph_fft_r8<true>(tid, 16, p, s, tw16[tid]));
        PHASE(ph_fft_r8<true>(tid, 128, p, s, tw128[tid]));
        PHASE(ph_fft_r8<true>(tid, 1024, p, s, tw1024[tid]));
    };
    auto round9 = [&]() {
        PHASE(ph9_r9<false>(tid, p, s));
        PHASE(ph9_r8<false>(tid, 64, p, s));
        PHASE(ph9_r8<false>(tid, 8, p, s));
        PHASE(ph9_mid(tid, p, s));
        PHASE(ph9_r8<true>(tid, 8, p, s));
        PHASE(ph9_r8<true>(tid, 64, p, s));
        PHASE(ph9_r9<true>(tid, p, s));
    };
    for (int blk = 0; blk < B * nb; ++blk) {
        std::memset(s, 0xff, smem.size());
        if (f9) {
            PHASE(ph9_load(tid, nt, blk, p, s));
            round9();
            PHASE(ph_hilb_mid<true>(tid, nt, blk, p, s));
            round9();
            PHASE(ph_hilb_env<true>(tid, nt, blk, p, s));
        } else {
            PHASE(ph_spec_load(tid, nt, blk, p, s));
            round();
            PHASE(ph_hilb_mid<false>(tid, nt, blk, p, s));
            round();
            PHASE(ph_hilb_env<false>(tid, nt, blk, p, s));
        }
        PHASE(ph_hilb_smooth(tid, nt, blk, p, s));
        PHASE(ph_hilb_dev(tid, nt, blk, p, s));
        PHASE(ph_hilb_finish(tid, nt, blk, p, s));
    }
#undef PHASE
    return 0;
}

// Spectrum of long frames (long_phases.h: packed real FFT through gfft.h, float32).
extern "C" int emul_long_spec(int N, int M, int K, int nb, const double* band, int B, float* spec_out) {
    LongParams p;
    std::memset(&p, 0, sizeof(p));
    if (!plan_long(p, N, M, K, nb, 0, 1) || !p.has_spec) return -2;
    std::vector<cpx<float>> roots, wn, uk, bw, bhat;
    long_spec_tables(p, roots, wn, uk, bw, bhat);
    p.sroots = roots.data(); p.swn = wn.data(); p.suk = uk.data(); p.sbw = bw.data(); p.sbhat = bhat.data();
    p.spec_db = 1; p.spec_in = band; p.spec_out = spec_out;
    std::vector<unsigned char> smem(p.spec_smem);
    unsigned char* s = smem.data();
    const int nt = kLSpecNT;
#define PHASE(...) for (int tid = 0; tid < nt; ++tid) { __VA_ARGS__; }
    for (int blk = 0; blk < B * nb; ++blk) {
        std::memset(s, 0xff, smem.size());
        if (p.spec_odd) {
#define SYNC
            ECHO_LSPEC_ODD_SEQUENCE(PHASE, SYNC, p, s, blk, nt);
#undef SYNC
            continue;
        }
        PHASE(ph_lspec_load(tid, nt, blk, p, s));
        int cur = 0;
        for (int ps = 0; ps < p.splan.npass; ++ps) { PHASE(ph_lspec_pass(tid, nt, p, s, ps, cur)); cur ^= 1; }
        PHASE(ph_lspec_store(tid, nt, blk, p, s, cur));
    }
#undef PHASE
    return 0;
}

// Spectrum of an odd-length long frame with the transform in place (ospec_phases.h).  Returns the cyclic length, or
// -2 when the frame length takes another kernel (even, or short enough for two buffers) or none.
#include "../../dsp_amd/csrc/ospec_phases.h"
extern "C" int emul_ospec(int N, int M, int K, int nb, const double* band, int B, int db, float* spec_out) {
    LongParams p;
    std::memset(&p, 0, sizeof(p));
    if (!plan_long(p, N, M, K, nb, 0, 1) || !plan_ospec(p)) return -2;
    std::vector<zf> tw, roots, bw, bhat;
    ospec_tables(p, tw, roots, bw, bhat);
    p.os_tw = tw.data(); p.os_roots = roots.data(); p.os_bhat = bhat.data(); p.sbw = bw.data();
    p.spec_db = db; p.spec_in = band; p.spec_out = spec_out;
    std::vector<unsigned char> smem(p.os_smem);
    unsigned char* s = smem.data();
    const int nt = kOsNT;
#define PHASE(...) for (int tid = 0; tid < nt; ++tid) { __VA_ARGS__; }
#define SYNC
    for (int blk = 0; blk < B * nb; ++blk) {
        std::memset(s, 0xff, smem.size());
        ECHO_OSPEC_SEQUENCE(PHASE, SYNC, p, s, blk, nt);
    }
#undef SYNC
#undef PHASE
    return p.os_m;
}

// 4608-point variant of the spectrum kernel (dsp_amd/csrc/fft9_phases.h).
extern "C" int emul_spec9(int N, int M, int K, int nb, const double* taps, const double* chirps, const double* band,
                          int B, float* spec_out) {
    BandParams p;
    std::memset(&p, 0, sizeof(p));
    if (!plan_band(p, N, M, K, nb, 0, 0, 0, 6.0) || !p.fft9) return -2;
    HostTables t;
    build_tables(t, p, taps, chirps);
    p.bw = t.bw.data(); p.post = t.post.data();
    p.f9tw9 = t.f9tw9.data(); p.f9twa = t.f9twa.data(); p.f9twb = t.f9twb.data(); p.f9bhat = t.f9bhat.data();
    p.spec_db = 1; p.specd_in = band; p.spec_out = spec_out;
    const int nt = kF9NT;
    std::vector<unsigned char> smem(p.spec_smem_bytes);
    unsigned char* s = smem.data();
#define PHASE(...) for (int tid = 0; tid < nt; ++tid) { __VA_ARGS__; }
    for (int blk = 0; blk < B * nb; ++blk) {
        std::memset(s, 0xff, smem.size());
        PHASE(ph9_load(tid, nt, blk, p, s));
        PHASE(ph9_r9<false>(tid, p, s));
        PHASE(ph9_r8<false>(tid, 64, p, s));
        PHASE(ph9_r8<false>(tid, 8, p, s));
        PHASE(ph9_mid(tid, p, s));
        PHASE(ph9_r8<true>(tid, 8, p, s));
        PHASE(ph9_r8<true>(tid, 64, p, s));
        PHASE(ph9_r9<true>(tid, p, s));
        PHASE(ph9_store(tid, nt, blk, p, s));
    }
#undef PHASE
    return 0;
}

// extern "C" entry points of libechohip.so (declared in include/echo_hip.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/echo_hip.h"
#include "echo_internal.h"
#include "tables.h"

using namespace echo;

constexpr int kEvPerSlot = 6;               // timing events per profiled chain call
constexpr int kInnerChunk = 8192;          // frames per pass when a many-pass call pipelines its own IIRs
constexpr int kSpecDftMax = 1 << 18;       // spectra by direct sums: n^2 / 2 complex multiply-adds per row

// Device scratch of one internal pass of the chain: everything beyond the caller's inputs and outputs.
struct Scratch {
    double* fwd = nullptr;                 // IIR forward-pass output
    double* y = nullptr;                   // IIR result
    float* bandf = nullptr;                // float32 band_sig handed from the band kernel to the spectrum kernel
    double* band = nullptr;                // long path: band_sig
    double* pos = nullptr;                 // long path: corr[half:] (or every lag)
    BandRes* res = nullptr;
    double* fused = nullptr;
    uint8_t* vany = nullptr;
    EnvRes* env = nullptr;                 // envelope-branch results, [frames][nb]
    B2Det* det = nullptr;                  // pair kernel -> exact kernel: float32 maxima and candidate lags, [frames][nb]
};

// Scratch of echo_chain_noalloc: one array per GPU (every device that loads the library holds its own copy at its own
// address until the process ends).  Host code never uses the array's name as an address: echo_create asks the runtime
// for the device address once per device and the entry point carves that pointer.
constexpr size_t kNoallocBytes = (size_t)ECHO_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kNoallocBytes];
constexpr int kMaxDevices = 64;
static unsigned char* g_noalloc_ptr[kMaxDevices];          // per device: device address of `scratch`, null until looked up

// Serialises the host side of the calls on one handle.  The reference calls its per-band functions from three pool
// threads at once (airSonar_optimized.py:717-719); a handle's scratch, plans, prefetch and timing state change inside a
// call, so every entry point that touches them holds this lock for its duration.  Recursive: echo_graph_create records an
// echo_chain call.  Copying a handle (lane clones) gives the copy a lock of its own.
struct HandleLock {
    std::recursive_mutex m;
    HandleLock() {}
    HandleLock(const HandleLock&) {}
    HandleLock& operator=(const HandleLock&) { return *this; }
};
#define ECHO_LOCK(h) std::lock_guard<std::recursive_mutex> echo_guard_((h)->lk.m)

struct EchoHandle {
    EchoConfig cfg;
    HandleLock lk;
    // The handle's scratch serves one call at a time on the device too: a call issued on another stream than the call
    // before it first makes its stream wait for everything queued on that earlier stream (order_scratch_use).
    hipStream_t last_stream = nullptr;
    bool last_stream_set = false;
    hipEvent_t xs_event = nullptr;
    hipEvent_t side_fork = nullptr, side_join = nullptr;   // ECHO_IIR_SIDE experiment
    int device = 0;
    bool bodd0 = false;
    BandParams base;                       // sizes, layout and device table pointers
    bool has_env = false;                  // envelope branch available (even correlation length with small prime factors)
    EnvParams ebase;
    cpx<double> *d_eroots = nullptr, *d_ewn = nullptr, *d_ewl = nullptr, *d_ebsw = nullptr, *d_ebsb = nullptr;
    cpx<double> *d_eltw = nullptr, *d_elwk = nullptr, *d_elwm = nullptr, *d_elgm = nullptr; double* d_elgam = nullptr;   // Hilbert transform as one real convolution
    int env_lin = 1;                       // ECHO_ENV_LIN=0: the envelope branch keeps the half-length transforms
    // full chain: FIR, correlation (real cyclic convolution) and envelope branch in one kernel (fce_phases.h);
    // ECHO_FCE=0 keeps band_kernel (direct sums for every lag) + corr_env_kernel
    cpx<double> *d_fcetw = nullptr, *d_fcewk = nullptr, *d_fceg = nullptr, *d_fcewm = nullptr, *d_fcegm = nullptr;
    bool fce_ok = false;
    double* d_corr = nullptr;              // full chain: raw correlation [frames][nb][L]
    double* d_sq = nullptr;                // full chain: signal_quality [frames][nb]
    float* d_spec = nullptr;               // full chain: dB spectra when the caller does not want them
    float2* d_uk = nullptr;
    cpx<float> *d_lroots = nullptr, *d_lwn = nullptr, *d_luk = nullptr;   // long path: spectrum FFT tables
    cpx<float> *d_lbw = nullptr, *d_lbhat = nullptr;                       // long path, odd N: chirp factors and filter spectrum
    double2* d_dftw = nullptr;             // long path, frame lengths no FFT kernel holds: the N roots of unity of the direct sums
    cpx<double> *d_cfg = nullptr, *d_cfw2 = nullptr, *d_cftw = nullptr;   // long path: FFT matched filter tables
    cpx<double> *d_rcwh = nullptr, *d_rcwk = nullptr, *d_rcroots = nullptr, *d_rcg = nullptr;   // the same as one real transform
    int* d_rcpos = nullptr;
    cpx<float> *d_ostw = nullptr, *d_osroots = nullptr, *d_osbhat = nullptr, *d_osbw = nullptr;   // odd-length spectrum in place
    // cross-batch pipelining: the IIR of the next batch on a stream of the handle's own (echo_chain_prefetch)
    hipStream_t pf_stream = nullptr;
    hipEvent_t pf_ready[2] = {nullptr, nullptr}, pf_consumed[2] = {nullptr, nullptr}, pf_t0 = nullptr, pf_t1 = nullptr;
    bool pf_consumed_set[2] = {false, false};
    hipEvent_t pf_rx_ready = nullptr;               // multi-pass calls: the caller's stream at call entry
    hipEvent_t chain_end[2] = {nullptr, nullptr};   // end of the last two chain calls (alternating)
    int chain_gen = 0; bool chain_end_set[2] = {false, false};
    hipEvent_t band_end = nullptr; bool band_end_set = false;   // behind the band stage of the latest chain call
    int pf_gate = 1;                                // 1: a prefetch starts when the chain call before the latest one has
                                                    // finished, i.e. it runs under the latest call's band kernel; 0: as soon as
                                                    // its buffer is free (ECHO_PF_GATE overrides, for experiments)
    double* pf_y[2] = {nullptr, nullptr};
    double* pf_fwd = nullptr;
    int pf_cap = 0, pf_next = 0;
    const float* pf_rx = nullptr;          // outstanding prefetch: input, size, slot
    int pf_B = 0, pf_slot = 0;
    bool pf_valid = false, pf_timed = false;
    bool long_path = false;                // frame too long for one workgroup, or the MAD detector: tiled kernels
    LongParams lbase;
    // device constant tables
    double *d_b = nullptr, *d_a = nullptr, *d_zi = nullptr, *d_taps_rev = nullptr, *d_chirp = nullptr;
    float2 *d_bw = nullptr, *d_bhat = nullptr, *d_tw = nullptr, *d_post = nullptr;
    float2 *d_f9tw9 = nullptr, *d_f9twa = nullptr, *d_f9twb = nullptr, *d_f9bhat = nullptr, *d_f9ghat = nullptr, *d_taps32x2 = nullptr;
    int use_band2 = 1;                     // ECHO_BAND2=0: the named chain keeps the single-frame band kernel
    // grow-only scratch, sized for cap_frames frames
    int cap_frames = 0;
    int cap_full = 0;                      // frames the full-chain scratch (d_corr, d_sq) is sized for
    // envelope plans for signals other than the correlation (echo_envelope_rows), made on first use
    struct EnvPlan { int L, smooth; EnvParams p; cpx<double> *roots, *wn, *wl, *bsw, *bsb, *ltw, *lwk, *lwm, *lgm; double* lgam; };
    std::vector<EnvPlan> env_plans;
    EnvRes* d_env_rows = nullptr; size_t env_rows_cap = 0;
    Scratch sc;                            // allocated by ensure_scratch
    unsigned char* noalloc = nullptr;      // device address of this device's static scratch array (echo_chain_noalloc)
    int max_chunk = 16384;                 // frames per internal pass of echo_chain
    int inner_pipeline = 1;                // ECHO_INNER_PIPELINE=0: calls of many passes run their IIRs in line
    int iir_mode = 0;                      // 0 = choose by batch size, 1 = lane per (frame, band), 2 = quad per (frame, band)
    int iir_lps = 0;                       // lanes per sequence in the quad kernel: 0 = choose, 4 or 16 (ECHO_IIR_LPS)
    int iir_prio = 0;                      // wave priority of the quad IIR kernel (ECHO_IIR_PRIO, experiments)
    // The chain's IIR on a stream shared by every handle of the device and confined to iir_cus compute units (0: all).
    // iir_side: 0 never, 1 always (ECHO_IIR_SIDE), 2 when the call is one pass of at least iir_side_min frames and a batch of
    // another handle is still in flight on the device (the default for long frames).
    int iir_side = 0, iir_cus = 0, iir_side_min = 2048;
    int iir_side_full = 0, iir_cus_full = 0;   // the same for echo_chain_full calls of short-frame handles
    long long side_passes = 0;             // passes whose IIR ran on that stream (echo_confined_iir_passes)
    bool counted = false;                  // included in the count of live handles (release_shared_streams)
    // profiling: a ring of event quads, one per profiled echo_chain call (first pass of the call)
    bool prof = false;
    std::vector<hipEvent_t> ev;            // 6 events per slot: start, after IIR, after band, after spectrum, end, after the pair kernel
    int prof_slots = 0, prof_used = 0;
    int prof_every = 1, prof_calls = 0;    // every prof_every-th armed call is timed (event records keep kernels of one stream apart)
    std::vector<char> prof_pair;           // per slot: the band stage ran as pair kernel + exact kernel (event 5 recorded)
    bool capturing = false;                // inside echo_graph_create's stream capture
    int live_graphs = 0;                   // recorded graphs hold pointers into the scratch: it must not move while > 0
    // Large calls split themselves into chunks that run on three streams of the handle's own (chain_lanes): lane 0 is
    // the handle itself, lanes 1 and 2 are clones that share its tables and own their scratch.
    bool is_clone = false;
    int fuse_det = 1;                      // ECHO_FUSE_DET=0: matched filter and median/MAD detector of long frames as two kernels
    int self_lanes = 1;                    // ECHO_LANES=0: large calls run pass after pass on the caller's stream
    bool lanes_ready = false;              // clones, streams and events of the lanes all exist
    EchoHandle* lane_clone[2] = {nullptr, nullptr};
    hipStream_t lane_stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t lane_fork = nullptr, lane_join[3] = {nullptr, nullptr, nullptr}, lane_kal[3] = {nullptr, nullptr, nullptr};
};

namespace {

// One stream per (device, number of compute units) on which the IIRs of every handle run one after the other, confined
// to that many compute units (0: all of them).  The mask takes the first n bits: the driver deals mask bits round-robin
// over the XCDs and their shader engines, so a multiple of 32 is an even share of every engine (others measured slower).
std::mutex g_side_mu;
struct SideStream { int ncu; hipStream_t st; };
std::vector<SideStream> g_side_streams[kMaxDevices];
struct InFlight { const void* who; hipEvent_t done; bool set; };
constexpr int kInFlight = 4;
InFlight g_inflight[kMaxDevices][kInFlight];
int g_inflight_next[kMaxDevices];
std::atomic<int> g_live_handles{0};

// The shared streams and events live as long as a handle does: the last echo_destroy releases them, and so does the
// end of the process for handles never destroyed (a compute-unit-masked stream still alive when the profiler's
// finaliser ran crashed rocprofv3 at exit).
void release_shared_streams_if(bool at_exit) {
    std::lock_guard<std::mutex> guard(g_side_mu);
    if (!at_exit && g_live_handles.load() != 0) return;      // a handle was created in the meantime
    for (int d = 0; d < kMaxDevices; ++d) {
        for (SideStream& e : g_side_streams[d]) { (void)hipStreamSynchronize(e.st); (void)hipStreamDestroy(e.st); }
        g_side_streams[d].clear();
        for (InFlight& e : g_inflight[d]) { if (e.done) (void)hipEventDestroy(e.done); e = InFlight{nullptr, nullptr, false}; }
        g_inflight_next[d] = 0;
    }
}

void release_shared_streams() { release_shared_streams_if(false); }

hipStream_t side_stream(int device, int ncu) {
    static std::once_flag at_exit;
    std::call_once(at_exit, [] { std::atexit([] { release_shared_streams_if(true); }); });
    std::lock_guard<std::mutex> guard(g_side_mu);
    for (const SideStream& e : g_side_streams[device]) if (e.ncu == ncu) return e.st;
    hipStream_t st = nullptr;
    if (ncu > 0) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < ncu && i < 256; ++i) mask[i / 32] |= 1u << (i % 32);
        if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) return nullptr;
    } else if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return nullptr;
    g_side_streams[device].push_back(SideStream{ncu, st});
    return st;
}

// Batches in flight: per device, an event at the end of the latest chain pass of each of the last few handles.  The
// events belong to this table, not to the handles, which come and go.  Guarded by g_side_mu.
bool others_in_flight(const EchoHandle* h) {
    std::lock_guard<std::mutex> guard(g_side_mu);
    for (const InFlight& e : g_inflight[h->device])
        if (e.set && e.who != h && hipEventQuery(e.done) == hipErrorNotReady) return true;
    return false;
}
void note_in_flight(const EchoHandle* h, hipStream_t s) {
    std::lock_guard<std::mutex> guard(g_side_mu);
    InFlight* slot = nullptr;
    for (InFlight& e : g_inflight[h->device]) if (e.who == h) slot = &e;
    if (!slot) { slot = &g_inflight[h->device][g_inflight_next[h->device]]; g_inflight_next[h->device] = (g_inflight_next[h->device] + 1) % kInFlight; }
    if (!slot->done && hipEventCreateWithFlags(&slot->done, hipEventDisableTiming) != hipSuccess) { slot->done = nullptr; slot->set = false; return; }
    slot->who = h;
    slot->set = hipEventRecord(slot->done, s) == hipSuccess;
}

template <class T> int dev_upload(T** dst, const T* src, size_t n) {
    if (hipMalloc(reinterpret_cast<void**>(dst), n * sizeof(T)) != hipSuccess) return ECHO_ERR_NOMEM;
    if (hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return ECHO_ERR_HIP;
    return ECHO_OK;
}

// hipFree waits for the device to go idle, so work still in flight on any stream never loses its buffers here.
void free_scratch(EchoHandle* h) {
    (void)hipFree(h->sc.fwd); (void)hipFree(h->sc.y); (void)hipFree(h->sc.res); (void)hipFree(h->sc.fused);
    (void)hipFree(h->sc.vany); (void)hipFree(h->sc.bandf); (void)hipFree(h->sc.band); (void)hipFree(h->sc.pos);
    (void)hipFree(h->sc.env); (void)hipFree(h->sc.det); (void)hipFree(h->d_corr); (void)hipFree(h->d_sq); (void)hipFree(h->d_spec);
    h->sc = Scratch();
    h->d_corr = nullptr; h->d_sq = nullptr; h->d_spec = nullptr;
    h->cap_full = 0;
    h->cap_frames = 0;
}

// Elements per frame-batch of each scratch buffer (shared by the allocating and the carving path).
struct ScratchSizes { size_t fwd, y, bandf, band, pos, res, fused, vany, env, det; };
ScratchSizes scratch_sizes(const EchoHandle* h, int frames) {
    const EchoConfig& c = h->cfg;
    const int P = 3 * c.iir_ncoef;
    const size_t ntiles = (size_t)(frames + 63) / 64;
    ScratchSizes z;
    z.fwd = (size_t)c.n_bands * ntiles * (size_t)(c.n_rx + 2 * P) * 64 * sizeof(double);
    z.y = (size_t)frames * c.n_bands * c.n_rx * sizeof(double);
    z.bandf = h->long_path ? 0 : (size_t)frames * c.n_bands * c.n_rx * sizeof(float);
    z.band = h->long_path ? z.y : 0;
    z.pos = h->long_path ? (size_t)frames * c.n_bands *
                           (size_t)(c.detector != ECHO_DET_MAD ? h->lbase.L : h->lbase.L - h->lbase.half) * sizeof(double) : 0;
    z.res = (size_t)frames * c.n_bands * sizeof(BandRes);
    z.fused = (size_t)frames * sizeof(double);
    z.vany = (size_t)frames;
    z.env = (size_t)frames * c.n_bands * sizeof(EnvRes);
    z.det = h->base.band2 ? (size_t)frames * c.n_bands * sizeof(B2Det) : 0;
    return z;
}

// Carves one pass's scratch out of a caller-provided device region; false when `bytes` is too small.
bool carve_scratch(const EchoHandle* h, int frames, unsigned char* base, size_t bytes, Scratch& sc) {
    const ScratchSizes z = scratch_sizes(h, frames);
    size_t off = 0;
    auto take = [&](size_t n) -> unsigned char* {
        if (n == 0) return nullptr;
        unsigned char* p = base + off;
        off += (n + 255) & ~(size_t)255;
        return p;
    };
    sc.fwd = reinterpret_cast<double*>(take(z.fwd));
    sc.y = reinterpret_cast<double*>(take(z.y));
    sc.bandf = reinterpret_cast<float*>(take(z.bandf));
    sc.band = reinterpret_cast<double*>(take(z.band));
    sc.pos = reinterpret_cast<double*>(take(z.pos));
    sc.res = reinterpret_cast<BandRes*>(take(z.res));
    sc.fused = reinterpret_cast<double*>(take(z.fused));
    sc.vany = reinterpret_cast<uint8_t*>(take(z.vany));
    sc.env = reinterpret_cast<EnvRes*>(take(z.env));
    sc.det = reinterpret_cast<B2Det*>(take(z.det));
    return off <= bytes;
}

int ensure_scratch(EchoHandle* h, int frames) {
    if (frames <= h->cap_frames) return ECHO_OK;
    if (h->live_graphs > 0) return ECHO_ERR_INVALID;       // a recorded graph still points into the current scratch
    free_scratch(h);
    const ScratchSizes z = scratch_sizes(h, frames);
    auto get = [](auto** p, size_t n) { return n == 0 || hipMalloc(reinterpret_cast<void**>(p), n) == hipSuccess; };
    if (!get(&h->sc.fwd, z.fwd) || !get(&h->sc.y, z.y) || !get(&h->sc.bandf, z.bandf) || !get(&h->sc.band, z.band) ||
        !get(&h->sc.pos, z.pos) || !get(&h->sc.res, z.res) || !get(&h->sc.fused, z.fused) ||
        !get(&h->sc.env, z.env) || !get(&h->sc.vany, z.vany) || !get(&h->sc.det, z.det)) {
        free_scratch(h);
        return ECHO_ERR_NOMEM;
    }
    h->cap_frames = frames;
    return ECHO_OK;
}

// scratch of the full chain (raw correlation, signal quality); allocated on first use
int ensure_full_scratch(EchoHandle* h, int frames) {
    if (frames <= h->cap_full) return ECHO_OK;
    if (h->capturing) return ECHO_ERR_INVALID;             // nothing may be allocated inside a stream capture
    (void)hipFree(h->d_corr); (void)hipFree(h->d_sq); (void)hipFree(h->d_spec);
    h->d_corr = nullptr; h->d_sq = nullptr; h->d_spec = nullptr; h->cap_full = 0;
    const size_t nbf = (size_t)frames * h->cfg.n_bands;
    if (hipMalloc(reinterpret_cast<void**>(&h->d_corr), nbf * (size_t)h->base.L * sizeof(double)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&h->d_sq), nbf * sizeof(double)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&h->d_spec), nbf * (size_t)h->base.nbins * sizeof(float)) != hipSuccess)
        return ECHO_ERR_NOMEM;
    h->cap_full = frames;
    return ECHO_OK;
}

// Zero-phase IIR for B frames.  Small batches cannot fill the chip with one lane per (frame, band) (a wave is a
// serial stream of ~370k instructions), so they run the quad-lane kernel: 4x the waves, each 1.8x shorter.
// corun: the IIR will run beside the arithmetic-bound kernels of another batch; from 8192 frames on the lane
// mapping (half the instruction issue of the quad mapping) is then the better neighbour.
int run_iir(EchoHandle* h, const float* rx, double* y, int B, hipStream_t s, double* fwd = nullptr, bool corun = false) {
    if (h->cfg.iir_ncoef == 1) return launch_iir_order0(rx, y, h->d_b, B, h->cfg.n_rx, h->cfg.n_bands, s);
    IirParams p;
    p.rx = rx; p.y = y; p.fwd = fwd ? fwd : h->sc.fwd; p.b = h->d_b; p.a = h->d_a; p.zi = h->d_zi;
    p.B = B; p.N = h->cfg.n_rx; p.P = 3 * h->cfg.iir_ncoef; p.nb = h->cfg.n_bands;
    p.y_aligned16 = (p.N % 2 == 0) && (reinterpret_cast<uintptr_t>(y) % 16 == 0);
    p.prio = h->iir_prio;
    const long lane_waves = (long)p.nb * ((B + 63) / 64);
    const bool quad = h->iir_mode == 2 || (h->iir_mode == 0 && lane_waves <= (corun ? 383 : 480));   // up to ~10k frames
    if (quad) {
        // a batch that leaves three quarters of the SIMDs idle in the quad mapping takes the row mapping: a quarter
        // of the frames per wave and half the dependent chain per sample (tick latency of the streaming mode)
        const long quad_waves = (long)p.nb * ((B + 15) / 16);
        const int lps = h->iir_lps ? h->iir_lps : ((quad_waves <= 192 && !corun) ? 16 : 4);
        p.ntiles = (B + 64 / lps - 1) / (64 / lps);
        return launch_iir_quad(p, h->cfg.iir_ncoef, lps, h->bodd0, s);
    }
    p.ntiles = (B + 63) / 64;
    return launch_iir(p, h->cfg.iir_ncoef, h->bodd0, s);
}

int ensure_prefetch(EchoHandle* h, int frames) {
    if (!h->pf_stream) {
        // Default priority on purpose.  A high-priority stream kept a hardware queue of its own even beside the
        // collective library's streams, but a process that also runs RCCL collectives then stopped making progress: the
        // prefetched IIR waits (in its high-priority queue) for events of the caller's normal-priority stream, which in
        // turn waits for a collective queued at normal priority, and that collective was never scheduled.  Callers that
        // want the prefetch stream on a queue of its own issue their first prefetch before they initialise RCCL.
        int pf_cus = 0;
        if (const char* m = std::getenv("ECHO_PF_CUS")) pf_cus = std::atoi(m);    // experiment: the prefetched IIR on a subset of the compute units
        if (pf_cus > 0) {
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < pf_cus && i < 256; ++i) mask[i / 32] |= 1u << (i % 32);
            if (hipExtStreamCreateWithCUMask(&h->pf_stream, 8, mask) != hipSuccess) return ECHO_ERR_HIP;
        } else if (hipStreamCreateWithFlags(&h->pf_stream, hipStreamNonBlocking) != hipSuccess) return ECHO_ERR_HIP;
        for (int i = 0; i < 2; ++i)
            if (hipEventCreateWithFlags(&h->pf_ready[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&h->pf_consumed[i], hipEventDisableTiming) != hipSuccess) return ECHO_ERR_HIP;
        if (hipEventCreate(&h->pf_t0) != hipSuccess || hipEventCreate(&h->pf_t1) != hipSuccess) return ECHO_ERR_HIP;
        if (hipEventCreateWithFlags(&h->pf_rx_ready, hipEventDisableTiming) != hipSuccess) return ECHO_ERR_HIP;
        for (int i = 0; i < 2; ++i)
            if (hipEventCreateWithFlags(&h->chain_end[i], hipEventDisableTiming) != hipSuccess) return ECHO_ERR_HIP;
        if (hipEventCreateWithFlags(&h->band_end, hipEventDisableTiming) != hipSuccess) return ECHO_ERR_HIP;
        if (const char* g = std::getenv("ECHO_PF_GATE")) h->pf_gate = std::atoi(g);
    }
    if (frames <= h->pf_cap) return ECHO_OK;
    if (hipStreamSynchronize(h->pf_stream) != hipSuccess) return ECHO_ERR_HIP;     // nothing may still use the old buffers
    (void)hipFree(h->pf_y[0]); (void)hipFree(h->pf_y[1]); (void)hipFree(h->pf_fwd);
    h->pf_y[0] = h->pf_y[1] = h->pf_fwd = nullptr; h->pf_cap = 0; h->pf_valid = false;
    const EchoConfig& c = h->cfg;
    const size_t ntiles = (size_t)(frames + 63) / 64;
    const size_t fwd_n = (size_t)c.n_bands * ntiles * (size_t)(c.n_rx + 6 * c.iir_ncoef) * 64;
    const size_t y_n = (size_t)frames * c.n_bands * c.n_rx;
    if (hipMalloc(reinterpret_cast<void**>(&h->pf_fwd), fwd_n * sizeof(double)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&h->pf_y[0]), y_n * sizeof(double)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&h->pf_y[1]), y_n * sizeof(double)) != hipSuccess)
        return ECHO_ERR_NOMEM;
    h->pf_cap = frames;
    return ECHO_OK;
}

// Starts the zero-phase IIR of rx[B][N] on the handle's stream into the free one of its two output buffers.
// after: an event the IIR must wait for (the producer of rx), or null; gate: honour the start rule of pf_gate.
int issue_prefetch(EchoHandle* h, const float* rx, int B, hipEvent_t after, bool gate) {
    const int slot = h->pf_next;
    // the previous result in this slot must have been read (band kernel of the pass that consumed it)
    if (h->pf_consumed_set[slot]) {
        if (hipStreamWaitEvent(h->pf_stream, h->pf_consumed[slot], 0) != hipSuccess) return ECHO_ERR_HIP;
        h->pf_consumed_set[slot] = false;
    }
    if (after && hipStreamWaitEvent(h->pf_stream, after, 0) != hipSuccess) return ECHO_ERR_HIP;
    if (gate && h->pf_gate == 2) {
        if (h->band_end_set) (void)hipStreamWaitEvent(h->pf_stream, h->band_end, 0);
    } else if (gate && h->pf_gate && h->chain_end_set[h->chain_gen ^ 1])
        (void)hipStreamWaitEvent(h->pf_stream, h->chain_end[h->chain_gen ^ 1], 0);
    (void)hipEventRecord(h->pf_t0, h->pf_stream);
    const int rc = run_iir(h, rx, h->pf_y[slot], B, h->pf_stream, h->pf_fwd, true);
    if (rc) return rc;
    (void)hipEventRecord(h->pf_t1, h->pf_stream);
    h->pf_timed = true;
    if (hipEventRecord(h->pf_ready[slot], h->pf_stream) != hipSuccess) return ECHO_ERR_HIP;
    h->pf_rx = rx; h->pf_B = B; h->pf_slot = slot; h->pf_valid = true;
    h->pf_next = slot ^ 1;
    return ECHO_OK;
}

FuseParams fuse_params(EchoHandle* h, const BandRes* res, EchoResult* out, double* fused, uint8_t* vany, int B) {
    FuseParams p;
    p.res = res; p.out = out; p.fused = fused; p.valid_any = vany;
    p.B = B; p.nb = h->cfg.n_bands; p.fs = h->cfg.fs; p.range_offset = h->cfg.range_offset;
    p.confidence = h->cfg.confidence; p.fusion = h->cfg.fusion; p.sound_speed = h->cfg.sound_speed;
    p.detector = h->cfg.detector;
    for (int j = 0; j < 3; ++j) p.fw[j] = h->cfg.freq_weights[j];
    p.env = nullptr; p.sq = nullptr; p.quality_out = nullptr;
    return p;
}

// The handle's tables, scratch and streams live on the device that was current in echo_create; a call issued with
// another device current would allocate and launch there.
bool on_device(const EchoHandle* h) {
    int dev = -1;
    return h && hipGetDevice(&dev) == hipSuccess && dev == h->device;
}

// A call that uses the handle's scratch on stream s while work of an earlier call may still use it on another stream:
// s waits for everything queued on that earlier stream so far (an event recorded there now covers the earlier call).
// Nothing is done inside a stream capture, and nothing when the stream is the one of the call before.
int order_scratch_use(EchoHandle* h, hipStream_t s) {
    if (h->capturing) return ECHO_OK;
    if (h->last_stream_set && h->last_stream != s) {
        if (!h->xs_event && hipEventCreateWithFlags(&h->xs_event, hipEventDisableTiming) != hipSuccess) return ECHO_ERR_HIP;
        if (hipEventRecord(h->xs_event, h->last_stream) == hipSuccess) {
            if (hipStreamWaitEvent(s, h->xs_event, 0) != hipSuccess) return ECHO_ERR_HIP;
        } else {
            // the earlier stream no longer exists; its work may still be running
            (void)hipGetLastError();
            if (hipDeviceSynchronize() != hipSuccess) return ECHO_ERR_HIP;
        }
    }
    h->last_stream = s; h->last_stream_set = true;
    return ECHO_OK;
}

bool detector_supported(const EchoHandle* h) {
    const int d = h->cfg.detector;
    return h->long_path ? (d >= ECHO_DET_MEDIAN && d <= ECHO_DET_TWOPEAK) : (d == ECHO_DET_MEDIAN || d == ECHO_DET_MEANPOW);
}

// spectra of long frames: the mixed-radix / chirp-z kernel where the frame length has a plan, direct sums otherwise
bool long_spec_available(const EchoHandle* h) { return h->lbase.has_spec || h->lbase.os_use || h->d_dftw; }
int run_long_spec(const EchoHandle* h, const LongParams& q, int rows_frames, hipStream_t s) {
    if (h->lbase.has_spec) return launch_long_spec(q, rows_frames, s);
    if (h->lbase.os_use) return launch_long_spec_odd_ip(q, rows_frames, s);
    if (!h->d_dftw) return ECHO_ERR_UNSUPPORTED;
    const int rc = launch_dft_rows(q.spec_in, h->d_dftw, q.spec_out, q.spec_db, rows_frames * q.nb, q.N, s);
    return rc == 0 ? ECHO_OK : ECHO_ERR_HIP;
}

}  // namespace

extern "C" {

const char* echo_version(void) { return "echohip 0.1 gfx950"; }

const char* echo_strerror(int code) {
    switch (code) {
        case ECHO_OK: return "ok";
        case ECHO_ERR_INVALID: return "invalid argument";
        case ECHO_ERR_UNSUPPORTED: return "configuration not supported by this build's kernels";
        case ECHO_ERR_NOMEM: return "device allocation failed";
        case ECHO_ERR_HIP: return "HIP runtime call or kernel launch failed";
        case ECHO_ERR_NODEVICE: return "no HIP device";
        case ECHO_ERR_TOO_LARGE: return "batch exceeds the fixed scratch of the entry point that allocates nothing";
        default: return "unknown error";
    }
}

int echo_create(const EchoConfig* cfg, EchoHandle** out) {
    if (!cfg || !out) return ECHO_ERR_INVALID;
    *out = nullptr;
    if (cfg->n_bands < 1 || cfg->n_bands > ECHO_MAX_BANDS || !cfg->b || !cfg->a || (!cfg->zi && cfg->iir_ncoef > 1) ||
        !cfg->taps || !cfg->chirps || cfg->n_rx < 64 || cfg->n_chirp < 1 || cfg->n_taps < 1 || cfg->fs <= 0)
        return ECHO_ERR_INVALID;
    // order 6 / 8 band-passes (13 / 17 coefficients), or order zero: the identity filter ([1], [1], [1]) design_filters
    // falls back to when a design fails (airSonar_optimized.py:178-182)
    if (cfg->iir_ncoef != 13 && cfg->iir_ncoef != 17 && cfg->iir_ncoef != 1) return ECHO_ERR_UNSUPPORTED;
    if (3 * cfg->iir_ncoef + 1 > 64 || 3 * cfg->iir_ncoef >= cfg->n_rx) return ECHO_ERR_UNSUPPORTED;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return ECHO_ERR_NODEVICE;
    EchoHandle* h = new (std::nothrow) EchoHandle();
    if (!h) return ECHO_ERR_NOMEM;
    h->cfg = *cfg;
    if (const char* m = std::getenv("ECHO_IIR_MODE")) h->iir_mode = std::atoi(m);   // development override
    if (const char* m = std::getenv("ECHO_INNER_PIPELINE")) h->inner_pipeline = std::atoi(m);
    if (const char* m = std::getenv("ECHO_LANES")) h->self_lanes = std::atoi(m);
    if (const char* m = std::getenv("ECHO_FUSE_DET")) h->fuse_det = std::atoi(m);
    if (const char* m = std::getenv("ECHO_IIR_PRIO")) h->iir_prio = std::atoi(m);
    if (const char* m = std::getenv("ECHO_IIR_LPS")) { const int v = std::atoi(m); h->iir_lps = (v == 4 || v == 16) ? v : 0; }
    (void)hipGetDevice(&h->device);
    // device address of this GPU's copy of the static scratch array, looked up once per device
    if (h->device < 0 || h->device >= kMaxDevices) { delete h; return ECHO_ERR_UNSUPPORTED; }
    if (!g_noalloc_ptr[h->device]) {
        unsigned char* ptr = nullptr;
        if (hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch)) != hipSuccess || !ptr) {
            delete h;
            return ECHO_ERR_HIP;
        }
        g_noalloc_ptr[h->device] = ptr;
    }
    h->noalloc = g_noalloc_ptr[h->device];
    std::memset(&h->base, 0, sizeof(h->base));
    const bool fits = plan_band(h->base, cfg->n_rx, cfg->n_chirp, cfg->n_taps, cfg->n_bands, cfg->min_delay,
                                cfg->noise_n, cfg->detector, cfg->snr_gate_db);
    h->long_path = !fits || cfg->detector == ECHO_DET_MAD || cfg->detector == ECHO_DET_TWOPEAK;
    std::memset(&h->lbase, 0, sizeof(h->lbase));
    if (h->long_path) {
        if (!plan_long(h->lbase, cfg->n_rx, cfg->n_chirp, cfg->n_taps, cfg->n_bands, cfg->min_delay, cfg->detector)) {
            delete h;
            return ECHO_ERR_UNSUPPORTED;
        }
        h->max_chunk = std::min(4096, 65535 / cfg->n_bands);
        h->lbase.noise_n = cfg->noise_n; h->lbase.snr_gate = cfg->snr_gate_db;
        {   // the noise window as NumPy's slices truncate it (airSonar_optimized.py:209, realtime_sonar.py:134)
            const int npos = h->lbase.L - h->lbase.half;
            int nn = cfg->noise_n;
            if (cfg->detector == ECHO_DET_MEDIAN && cfg->min_delay + nn > npos) nn = npos > cfg->min_delay ? npos - cfg->min_delay : 0;
            if (cfg->detector == ECHO_DET_MEANPOW && nn > npos) nn = npos;
            if (cfg->detector == ECHO_DET_MEDIAN || cfg->detector == ECHO_DET_MEANPOW) h->lbase.noise_n = nn;
        }
    }
    if (h->long_path) {
        // Every other kernel of the long-frame chain fills the LDS of a compute unit, so a unit that holds an IIR
        // workgroup (51 KB, one wave per SIMD, 1.1 ms of a latency-bound recurrence) is closed to them: batches in flight
        // on several streams then overlap the IIR with nothing, it merely holds 192 of the 256 units.  Confined to half of
        // the units its workgroups stack (two or three per unit, the recurrence hardly slower) and the other half stays
        // with the matched filter, FIR and spectrum kernels of the other batches: airSonar.py's set 1.21 -> 1.34 M
        // frames/s, 666.py's 1.30 -> 1.44 M under three lanes.
        int ncu = 0;
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device);
        // A batch alone on the device gains nothing from it (its IIR has nothing to make room for and runs 1.8 instead of
        // 1.1 ms), and small batches lose to the two extra event hops: hence the in-flight test and the size floor.
        h->iir_cus = (ncu / 2) / 32 * 32;
        h->iir_side = h->iir_cus > 0 ? 2 : 0;
    }
    else {
        // The full chain's envelope kernel fills a unit's LDS too (two buffers of 4341 complex doubles); its IIR is the
        // shorter order-6 one, and an eighth of the units (one per shader engine) is the share that paid: 1.02 -> 1.10 M
        // frames/s under three lanes (a quarter and more: no gain).
        int ncu = 0;
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device);
        h->iir_cus_full = (ncu / 8) / 32 * 32;
        h->iir_side_full = h->iir_cus_full > 0 ? 2 : 0;
    }
    if (const char* m = std::getenv("ECHO_IIR_SIDE")) h->iir_side = h->iir_side_full = std::atoi(m) ? 1 : 0;   // overrides, experiments
    if (const char* m = std::getenv("ECHO_IIR_CUS")) h->iir_cus = h->iir_cus_full = std::max(0, std::atoi(m));
    if (const char* m = std::getenv("ECHO_IIR_SIDE_MIN")) h->iir_side_min = std::max(0, std::atoi(m));
    const int nb = cfg->n_bands, nc = cfg->iir_ncoef;
    h->bodd0 = true;
    for (int j = 0; j < nb; ++j)
        for (int k = 1; k < nc; k += 2)
            if (cfg->b[j * nc + k] != 0.0) h->bodd0 = false;
    for (int j = 0; j < nb; ++j)
        if (cfg->a[j * nc] != 1.0) { delete h; return ECHO_ERR_INVALID; }   // tables must be a0-normalised
    HostTables t;
    if (const char* m = std::getenv("ECHO_BAND2")) h->use_band2 = std::atoi(m);       // development override
    if (!h->base.fft9 || !h->use_band2) h->base.band2 = 0;
    if (h->long_path) build_coef_tables(t, nb, cfg->n_taps, cfg->n_chirp, cfg->taps, cfg->chirps);
    else {
        build_tables(t, h->base, cfg->taps, cfg->chirps);
        if (h->base.band2) band2_bounds(h->base, cfg->taps, cfg->chirps);
    }
    int rc = ECHO_OK;
    auto up = [&](int r) { if (rc == ECHO_OK) rc = r; };
    up(dev_upload(&h->d_b, cfg->b, (size_t)nb * nc));
    up(dev_upload(&h->d_a, cfg->a, (size_t)nb * nc));
    if (nc > 1) up(dev_upload(&h->d_zi, cfg->zi, (size_t)nb * (nc - 1)));
    up(dev_upload(&h->d_taps_rev, t.taps_rev.data(), t.taps_rev.size()));
    up(dev_upload(&h->d_chirp, t.chirp.data(), t.chirp.size()));
    if (!h->long_path) {
        up(dev_upload(&h->d_bw, t.bw.data(), t.bw.size()));
        up(dev_upload(&h->d_bhat, t.bhat.data(), t.bhat.size()));
        up(dev_upload(&h->d_tw, t.tw.data(), t.tw.size()));
        up(dev_upload(&h->d_post, t.post.data(), t.post.size()));
        up(dev_upload(&h->d_uk, t.uk.data(), t.uk.size()));
        if (h->base.fft9) {
            up(dev_upload(&h->d_f9tw9, t.f9tw9.data(), t.f9tw9.size()));
            up(dev_upload(&h->d_f9twa, t.f9twa.data(), t.f9twa.size()));
            up(dev_upload(&h->d_f9twb, t.f9twb.data(), t.f9twb.size()));
            up(dev_upload(&h->d_f9bhat, t.f9bhat.data(), t.f9bhat.size()));
        }
        if (h->base.band2) {
            up(dev_upload(&h->d_f9ghat, t.f9ghat.data(), t.f9ghat.size()));
            up(dev_upload(&h->d_taps32x2, t.taps32x2.data(), t.taps32x2.size()));
            if (rc == ECHO_OK) rc = band2_kernel_prepare(h->base.b2_smem_bytes, h->base.b2_exact_smem_bytes);
        }
        if (const char* pad = std::getenv("ECHO_BAND_SMEM_PAD")) h->base.smem_bytes += std::atoi(pad);   // development: occupancy experiments
        if (const char* f9 = std::getenv("ECHO_FFT9")) h->base.fft9 = h->base.fft9 && std::atoi(f9) != 0;   // development override
        if (!h->base.fft9) h->base.spec_smem_bytes = kFftM * 8;
        if (rc == ECHO_OK)
            rc = band_kernel_prepare(h->base.smem_bytes, h->base.spec_smem_bytes, h->base.spec_hilb_smem_bytes);
    } else if (rc == ECHO_OK) {
        if (h->lbase.has_spec) {
            std::vector<cpx<float>> lr, lw, lu, lbw, lbh;
            long_spec_tables(h->lbase, lr, lw, lu, lbw, lbh);
            up(dev_upload(&h->d_lroots, lr.data(), lr.size()));
            up(dev_upload(&h->d_lwn, lw.data(), lw.size()));
            up(dev_upload(&h->d_luk, lu.data(), lu.size()));
            h->lbase.sroots = h->d_lroots; h->lbase.swn = h->d_lwn; h->lbase.suk = h->d_luk;
            if (h->lbase.spec_odd) {
                up(dev_upload(&h->d_lbw, lbw.data(), lbw.size()));
                up(dev_upload(&h->d_lbhat, lbh.data(), lbh.size()));
                h->lbase.sbw = h->d_lbw; h->lbase.sbhat = h->d_lbhat;
            }
        }
        if (!h->lbase.has_spec && plan_ospec(h->lbase)) {
            // an odd frame length beyond the two-buffer chirp-z kernel: the same convolution with the transform in place
            std::vector<zf> tw, rr, bw, bh;
            ospec_tables(h->lbase, tw, rr, bw, bh);
            up(dev_upload(&h->d_ostw, tw.data(), tw.size()));
            up(dev_upload(&h->d_osroots, rr.data(), rr.size()));
            up(dev_upload(&h->d_osbw, bw.data(), bw.size()));
            up(dev_upload(&h->d_osbhat, bh.data(), bh.size()));
            h->lbase.os_tw = h->d_ostw; h->lbase.os_roots = h->d_osroots; h->lbase.os_bhat = h->d_osbhat; h->lbase.sbw = h->d_osbw;
            if (rc == ECHO_OK) rc = ospec_kernel_prepare(h->lbase);
        }
        if (!h->lbase.has_spec && !h->lbase.os_use && cfg->n_rx <= kSpecDftMax) {
            // a frame length none of the FFT kernels takes (e.g. an odd N beyond one workgroup's LDS): spectra as direct
            // float64 sums, O(N^2) per frame and band
            std::vector<double2> w((size_t)cfg->n_rx);
            const long double tau = 6.283185307179586476925286766559005768L;
            for (int t = 0; t < cfg->n_rx; ++t) {
                const long double a = -tau * (long double)t / (long double)cfg->n_rx;
                w[t] = double2{(double)cosl(a), (double)sinl(a)};
            }
            up(dev_upload(&h->d_dftw, w.data(), w.size()));
        }
        if (const char* st = std::getenv("ECHO_DET_STAGES")) h->lbase.det_stages = std::atoi(st);   // development aids
        if (const char* st = std::getenv("ECHO_DET_PASSES")) h->lbase.det_passes = std::atoi(st);
        if (const char* cf = std::getenv("ECHO_CORR_FFT")) h->lbase.cf_use = h->lbase.cf_use && std::atoi(cf) != 0;   // development override
        if (h->lbase.cf_use) {
            std::vector<zd> g, w2, tw;
            cf_tables(nb, cfg->n_chirp, cfg->chirps, g, w2, tw);
            up(dev_upload(&h->d_cfg, g.data(), g.size()));
            up(dev_upload(&h->d_cfw2, w2.data(), w2.size()));
            up(dev_upload(&h->d_cftw, tw.data(), tw.size()));
            h->lbase.cf_g = h->d_cfg; h->lbase.cf_w2 = h->d_cfw2; h->lbase.cf_tw = h->d_cftw;
            if (rc == ECHO_OK) rc = cfft_kernel_prepare();
            // ECHO_RCF=1: the detectors that read one contiguous lag range take the matched filter as a single real
            // transform (rcf_phases.h).  Fewer points and 50 KB less LDS, yet slower on gfx950 than the power-of-two
            // transform above (its radix-10/9/6 passes cost more than they save), so it is not the default.
            const char* rcf = std::getenv("ECHO_RCF");
            const int det = cfg->detector;
            if (rcf && std::atoi(rcf) != 0 && (det == ECHO_DET_MAD || det == ECHO_DET_TWOPEAK) &&
                (det != ECHO_DET_MAD || h->lbase.L - h->lbase.half > h->lbase.min_delay) &&
                plan_rcf(h->lbase, det == ECHO_DET_MAD ? h->lbase.half + h->lbase.min_delay : 0, std::atoi(rcf) == 2)) {
                std::vector<zd> wh, wk, rr, rg;
                std::vector<int> pos;
                rcf_tables(h->lbase, nb, cfg->chirps, wh, wk, rr, rg, pos);
                up(dev_upload(&h->d_rcwh, wh.data(), wh.size()));
                up(dev_upload(&h->d_rcwk, wk.data(), wk.size()));
                up(dev_upload(&h->d_rcroots, rr.data(), rr.size()));
                up(dev_upload(&h->d_rcg, rg.data(), rg.size()));
                up(dev_upload(&h->d_rcpos, pos.data(), pos.size()));
                h->lbase.rc_wh = h->d_rcwh; h->lbase.rc_wk = h->d_rcwk; h->lbase.rc_roots = h->d_rcroots;
                h->lbase.rc_g = h->d_rcg; h->lbase.rc_pos = h->d_rcpos;
                if (rc == ECHO_OK) rc = rcf_kernel_prepare(h->lbase);
            }
        }
        if (rc == ECHO_OK) rc = long_kernels_prepare(h->lbase);
    }
    if (rc == ECHO_OK && nc > 1 && iir_kernel_prepare() != 0) rc = ECHO_ERR_HIP;
    std::memset(&h->ebase, 0, sizeof(h->ebase));
    if (const char* el = std::getenv("ECHO_ENV_LIN")) h->env_lin = std::atoi(el);
    if (rc == ECHO_OK && !h->long_path &&
        plan_env(h->ebase, cfg->n_rx + cfg->n_chirp - 1, nb, cfg->min_delay, 5, h->env_lin != 0)) {
        std::vector<cpx<double>> er, ewn, ewl;
        if (h->ebase.lin) {
            std::vector<double> gam;
            envlin_tables(h->ebase.L, er, ewn, gam);
            up(dev_upload(&h->d_eltw, er.data(), er.size()));
            up(dev_upload(&h->d_elwk, ewn.data(), ewn.size()));
            up(dev_upload(&h->d_elgam, gam.data(), gam.size()));
            std::vector<cpx<double>> wm, gm;
            envlin_mid_tables(ewn, gam, wm, gm);
            up(dev_upload(&h->d_elwm, wm.data(), wm.size()));
            up(dev_upload(&h->d_elgm, gm.data(), gm.size()));
            h->ebase.lin_tw = h->d_eltw; h->ebase.lin_wk = h->d_elwk; h->ebase.lin_gam = h->d_elgam;
            h->ebase.lin_wmid = h->d_elwm; h->ebase.lin_gmid = h->d_elgm;
        } else {
        env_tables(h->ebase, er, ewn, ewl);
        up(dev_upload(&h->d_eroots, er.data(), er.size()));
        up(dev_upload(&h->d_ewn, ewn.data(), ewn.size()));
        up(dev_upload(&h->d_ewl, ewl.data(), ewl.size()));
        }
        if (h->ebase.bs_m) {
            // half the correlation length has a prime factor beyond the mixed-radix passes (e.g. L = 4562 = 2 * 2281):
            // the Hilbert transform's half-length FFTs run as Bluestein convolutions, as in echo_envelope_rows
            std::vector<cpx<double>> bw, bb;
            env_bs_tables(h->ebase, bw, bb);
            up(dev_upload(&h->d_ebsw, bw.data(), bw.size()));
            up(dev_upload(&h->d_ebsb, bb.data(), bb.size()));
            h->ebase.bs_w = h->d_ebsw; h->ebase.bs_bhat = h->d_ebsb;
        }
        if (rc == ECHO_OK) rc = env_kernel_prepare(h->ebase);
        const char* fce_env = std::getenv("ECHO_FCE");
        if (rc == ECHO_OK && !(fce_env && std::atoi(fce_env) == 0) && cfg->iir_ncoef > 1 &&
            plan_fce(h->ebase, cfg->n_rx, cfg->n_chirp, cfg->n_taps)) {
            std::vector<cpx<double>> ftw, fwk, fg;
            fce_tables(cfg->n_chirp, nb, cfg->chirps, ftw, fwk, fg);
            up(dev_upload(&h->d_fcetw, ftw.data(), ftw.size()));
            up(dev_upload(&h->d_fcewk, fwk.data(), fwk.size()));
            up(dev_upload(&h->d_fceg, fg.data(), fg.size()));
            std::vector<cpx<double>> fwm, fgm;
            fce_mid_tables(nb, fwk, fg, fwm, fgm);
            up(dev_upload(&h->d_fcewm, fwm.data(), fwm.size()));
            up(dev_upload(&h->d_fcegm, fgm.data(), fgm.size()));
            h->ebase.fce_tw = h->d_fcetw; h->ebase.fce_wk = h->d_fcewk; h->ebase.fce_g = h->d_fceg;
            h->ebase.fce_wmid = h->d_fcewm; h->ebase.fce_gmid = h->d_fcegm;
            h->ebase.taps_rev = h->d_taps_rev;
            if (rc == ECHO_OK) rc = fce_kernel_prepare(h->ebase);
            h->fce_ok = rc == ECHO_OK;
            // two workgroups of the fused kernel fill a compute unit's LDS whatever else runs: moving the IIR to the
            // confined stream by itself no longer pays for this chain (1.73 M against 1.76 M frames/s in line)
            if (h->fce_ok && h->iir_side_full == 2) h->iir_side_full = 0;
        }
        if (const char* st = std::getenv("ECHO_ENV_STOP")) h->ebase.stop_after = std::atoi(st);   // development aid
        h->ebase.roots = h->d_eroots; h->ebase.wn = h->d_ewn; h->ebase.wl = h->d_ewl;
        h->has_env = rc == ECHO_OK;
    }
    if (rc != ECHO_OK) { echo_destroy(h); return rc; }
    h->base.taps_rev = h->d_taps_rev; h->base.chirp = h->d_chirp;
    h->base.bw = h->d_bw; h->base.bhat = h->d_bhat; h->base.tw = h->d_tw; h->base.post = h->d_post;
    h->base.uk = h->d_uk;
    h->base.f9tw9 = h->d_f9tw9; h->base.f9twa = h->d_f9twa; h->base.f9twb = h->d_f9twb; h->base.f9bhat = h->d_f9bhat;
    h->base.f9ghat = h->d_f9ghat; h->base.taps32x2 = h->d_taps32x2;
    h->lbase.taps_rev = h->d_taps_rev; h->lbase.chirp = h->d_chirp;
    // the cfg table pointers are host pointers owned by the caller; do not keep them
    h->cfg.b = h->cfg.a = h->cfg.zi = h->cfg.taps = h->cfg.chirps = nullptr;
    h->counted = true;
    g_live_handles.fetch_add(1);
    *out = h;
    return ECHO_OK;
}

void echo_destroy(EchoHandle* h) {
    if (!h) return;
    for (EchoHandle*& c : h->lane_clone) {
        if (!c) continue;
        free_scratch(c);
        if (c->side_fork) (void)hipEventDestroy(c->side_fork);
        if (c->side_join) (void)hipEventDestroy(c->side_join);
        if (c->xs_event) (void)hipEventDestroy(c->xs_event);
        delete c; c = nullptr;
    }
    for (hipStream_t& st : h->lane_stream) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); st = nullptr; }
    if (h->lane_fork) (void)hipEventDestroy(h->lane_fork);
    for (hipEvent_t e : h->lane_join) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->lane_kal) if (e) (void)hipEventDestroy(e);
    if (h->xs_event) (void)hipEventDestroy(h->xs_event);
    if (h->side_fork) (void)hipEventDestroy(h->side_fork);
    if (h->side_join) (void)hipEventDestroy(h->side_join);
    free_scratch(h);
    void* tabs[] = {h->d_b, h->d_a, h->d_zi, h->d_taps_rev, h->d_chirp, h->d_bw, h->d_bhat, h->d_tw, h->d_post,
                    h->d_eroots, h->d_ewn, h->d_ewl, h->d_ebsw, h->d_ebsb, h->d_uk, h->d_lroots, h->d_lwn, h->d_luk, h->d_lbw, h->d_lbhat,
                    h->d_f9tw9, h->d_f9twa, h->d_f9twb, h->d_f9bhat, h->d_f9ghat, h->d_taps32x2, h->d_cfg, h->d_cfw2, h->d_cftw, h->d_dftw,
                    h->d_rcwh, h->d_rcwk, h->d_rcroots, h->d_rcg, h->d_rcpos,
                    h->d_ostw, h->d_osroots, h->d_osbhat, h->d_osbw, h->d_eltw, h->d_elwk, h->d_elgam,
                    h->d_fcetw, h->d_fcewk, h->d_fceg, h->d_fcewm, h->d_fcegm, h->d_elwm, h->d_elgm};
    for (void* t : tabs) (void)hipFree(t);
    for (auto& e : h->ev) if (e) (void)hipEventDestroy(e);
    for (auto& pl : h->env_plans) {
        (void)hipFree(pl.roots); (void)hipFree(pl.wn); (void)hipFree(pl.wl); (void)hipFree(pl.bsw); (void)hipFree(pl.bsb);
        (void)hipFree(pl.ltw); (void)hipFree(pl.lwk); (void)hipFree(pl.lgam); (void)hipFree(pl.lwm); (void)hipFree(pl.lgm);
    }
    (void)hipFree(h->d_env_rows);
    if (h->pf_stream) {
        (void)hipStreamSynchronize(h->pf_stream);
        (void)hipStreamDestroy(h->pf_stream);
        hipEvent_t evs[] = {h->pf_ready[0], h->pf_ready[1], h->pf_consumed[0], h->pf_consumed[1], h->pf_t0, h->pf_t1,
                            h->chain_end[0], h->chain_end[1], h->pf_rx_ready, h->band_end};
        for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    }
    (void)hipFree(h->pf_y[0]); (void)hipFree(h->pf_y[1]); (void)hipFree(h->pf_fwd);
    const bool last = h->counted && g_live_handles.fetch_sub(1) == 1;
    delete h;
    if (last) release_shared_streams();
}

int echo_reserve(EchoHandle* h, int max_frames) {
    if (h && !on_device(h)) return ECHO_ERR_INVALID;         // the handle belongs to another device
    if (!h || max_frames < 1) return ECHO_ERR_INVALID;
    ECHO_LOCK(h);
    return ensure_scratch(h, std::min(max_frames, h->max_chunk));
}

int echo_set_profiling_every(EchoHandle* h, int slots, int every) {
    if (every < 1 || !h) return ECHO_ERR_INVALID;
    ECHO_LOCK(h);
    const int rc = echo_set_profiling(h, slots);
    if (rc == ECHO_OK) h->prof_every = every;
    return rc;
}

int echo_set_profiling(EchoHandle* h, int slots) {
    if (!h || slots < 0) return ECHO_ERR_INVALID;
    ECHO_LOCK(h);
    h->prof = slots > 0;
    h->prof_used = 0; h->prof_every = 1; h->prof_calls = 0;
    while ((int)h->ev.size() < kEvPerSlot * slots) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) return ECHO_ERR_HIP;   // timing only: no cache flush behind a kernel
        h->ev.push_back(e);
    }
    h->prof_slots = slots;
    return ECHO_OK;
}

int echo_last_kernel_ms(EchoHandle* h, float* iir_ms, float* band_ms, float* spec_ms, float* fuse_ms) {
    if (!h) return ECHO_ERR_INVALID;
    ECHO_LOCK(h);
    if (!h->prof || h->prof_used < 1) return ECHO_ERR_INVALID;
    double acc[4] = {0, 0, 0, 0};
    for (int k = 0; k < h->prof_used; ++k) {
        hipEvent_t* e = &h->ev[kEvPerSlot * k];
        if (hipEventSynchronize(e[4]) != hipSuccess) return ECHO_ERR_HIP;
        for (int j = 0; j < 4; ++j) {
            float x = 0;
            if (hipEventElapsedTime(&x, e[j], e[j + 1]) != hipSuccess) return ECHO_ERR_HIP;
            acc[j] += x;
        }
    }
    const double n = (double)h->prof_used;
    if (iir_ms) *iir_ms = (float)(acc[0] / n);
    if (band_ms) *band_ms = (float)(acc[1] / n);
    if (spec_ms) *spec_ms = (float)(acc[2] / n);
    if (fuse_ms) *fuse_ms = (float)(acc[3] / n);
    h->prof_used = 0;
    return ECHO_OK;
}

long long echo_confined_iir_passes(EchoHandle* h) {
    if (!h) return -1;
    ECHO_LOCK(h);
    long long n = h->side_passes;
    for (const EchoHandle* c : h->lane_clone) if (c) n += c->side_passes;
    return n;
}

int echo_kernel_times(EchoHandle* h, float* ms, int n) {
    if (!h || !ms || n < 1) return ECHO_ERR_INVALID;
    ECHO_LOCK(h);
    if (!h->prof || h->prof_used < 1) return ECHO_ERR_INVALID;
    double acc[6] = {0, 0, 0, 0, 0, 0};              // iir, band stage, spectrum, fusion, pair kernel, exact kernel
    int npair = 0;
    for (int k = 0; k < h->prof_used; ++k) {
        hipEvent_t* e = &h->ev[kEvPerSlot * k];
        if (hipEventSynchronize(e[4]) != hipSuccess) return ECHO_ERR_HIP;
        for (int j = 0; j < 4; ++j) {
            float x = 0;
            if (hipEventElapsedTime(&x, e[j], e[j + 1]) != hipSuccess) return ECHO_ERR_HIP;
            acc[j] += x;
        }
        if (k < (int)h->prof_pair.size() && h->prof_pair[k]) {
            float a = 0, b = 0;
            if (hipEventElapsedTime(&a, e[1], e[5]) != hipSuccess || hipEventElapsedTime(&b, e[5], e[2]) != hipSuccess)
                return ECHO_ERR_HIP;
            acc[4] += a; acc[5] += b; ++npair;
        }
    }
    const double cnt = (double)h->prof_used;
    for (int j = 0; j < n && j < 6; ++j) ms[j] = (float)(j < 4 ? acc[j] / cnt : (npair ? acc[j] / npair : 0.0));
    for (int j = 6; j < n; ++j) ms[j] = 0.f;
    h->prof_used = 0;
    return ECHO_OK;
}

int echo_fir_bank(EchoHandle* h, const float* rx, double* out, int B, void* stream) {
    if (h && !on_device(h)) return ECHO_ERR_INVALID;         // the handle belongs to another device
    if (h && B == 0) return ECHO_OK;                       // empty batch: nothing to do
    if (!h || !rx || !out || B < 0) return ECHO_ERR_INVALID;
    if (h->long_path) {
        for (int f0 = 0; f0 < B; f0 += h->max_chunk) {
            LongParams q = h->lbase;
            q.fir_rx = 1; q.rx_in = rx + (size_t)f0 * q.N; q.band_out = out + (size_t)f0 * q.nb * q.N;
            if (int rc = launch_long_fir(q, std::min(h->max_chunk, B - f0), static_cast<hipStream_t>(stream))) return rc;
        }
        return ECHO_OK;
    }
    BandParams p = h->base;
    p.mode = MODE_RX_FIR; p.do_corr = 0; p.rx_in = rx; p.band_out = out;
    return launch_band(p, B, static_cast<hipStream_t>(stream));
}

int echo_bandpass_bank(EchoHandle* h, const float* rx, double* band_sig, int B, void* stream) {
    if (h && !on_device(h)) return ECHO_ERR_INVALID;         // the handle belongs to another device
    if (h && B == 0) return ECHO_OK;                       // empty batch: nothing to do
    if (!h || !rx || !band_sig || B < 0) return ECHO_ERR_INVALID;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ECHO_LOCK(h);
    if (int rc = order_scratch_use(h, s)) return rc;
    for (int f0 = 0; f0 < B; f0 += h->max_chunk) {
        const int n = std::min(h->max_chunk, B - f0);
        int rc = ensure_scratch(h, n);
        if (rc) return rc;
        const size_t o = (size_t)f0 * h->cfg.n_bands * h->cfg.n_rx;
        rc = run_iir(h, rx + (size_t)f0 * h->cfg.n_rx, h->sc.y, n, s);
        if (rc) return rc;
        if (h->long_path) {
            LongParams q = h->lbase;
            q.fir_rx = 0; q.d_in = h->sc.y; q.band_out = band_sig + o;
            rc = launch_long_fir(q, n, s);
            if (rc) return rc;
            continue;
        }
        BandParams p = h->base;
        p.mode = MODE_Y_FIR; p.do_corr = 0; p.d_in = h->sc.y; p.band_out = band_sig + o;
        rc = launch_band(p, n, s);
        if (rc) return rc;
    }
    return ECHO_OK;
}

int echo_spectrum(EchoHandle* h, const double* band_sig, float* spec, int db, int B, void* stream) {
    if (h && !on_device(h)) return ECHO_ERR_INVALID;         // the handle belongs to another device
    if (h && B == 0) return ECHO_OK;                       // empty batch: nothing to do
    if (!h || !band_sig || !spec || B < 0) return ECHO_ERR_INVALID;
    if (h->long_path) {
        for (int f0 = 0; f0 < B; f0 += h->max_chunk) {
            LongParams q = h->lbase;
            q.spec_db = db ? 1 : 0;
            q.spec_in = band_sig + (size_t)f0 * q.nb * q.N; q.spec_out = spec + (size_t)f0 * q.nb * q.nbins;
            if (int rc = run_long_spec(h, q, std::min(h->max_chunk, B - f0), static_cast<hipStream_t>(stream))) return rc;
        }
        return ECHO_OK;
    }
    BandParams p = h->base;
    p.spec_db = db ? 1 : 0; p.specd_in = band_sig; p.spec_out = spec;
    return launch_spec(p, B, static_cast<hipStream_t>(stream));
}

int echo_mag2db(const float* x, float* out, size_t n, void* stream) {
    if ((!x || !out) && n) return ECHO_ERR_INVALID;
    return launch_mag2db(x, out, n, static_cast<hipStream_t>(stream));
}

int echo_correlate(EchoHandle* h, const double* band_sig, double* corr, int B, void* stream) {
    if (h && !on_device(h)) return ECHO_ERR_INVALID;         // the handle belongs to another device
    if (h && B == 0) return ECHO_OK;                       // empty batch: nothing to do
    if (!h || !band_sig || !corr || B < 0) return ECHO_ERR_INVALID;
    if (h->long_path) {
        for (int f0 = 0; f0 < B; f0 += h->max_chunk) {
            LongParams q = h->lbase;
            q.band_in = band_sig + (size_t)f0 * q.nb * q.N; q.corr_out = corr + (size_t)f0 * q.nb * q.L;
            q.k_begin = 0; q.corr_k0 = 0; q.corr_stride = q.L;
            if (int rc = launch_long_corr(q, std::min(h->max_chunk, B - f0), static_cast<hipStream_t>(stream))) return rc;
        }
        return ECHO_OK;
    }
    BandParams p = h->base;
    p.mode = MODE_BAND; p.do_corr = 1; p.d_in = band_sig; p.corr_io = corr; p.res_out = nullptr;
    return launch_band(p, B, static_cast<hipStream_t>(stream));
}

int echo_first_strong_peak(EchoHandle* h, double* corr, int32_t* peak_idx, double* snr_db, double* amplitude,
                           int B, void* stream) {
    if (h && !on_device(h)) return ECHO_ERR_INVALID;         // the handle belongs to another device
    if (h && B == 0) return ECHO_OK;                       // empty batch: nothing to do
    if (!h || !corr || !peak_idx || !snr_db || B < 0) return ECHO_ERR_INVALID;
    if (!detector_supported(h)) return ECHO_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nb = h->cfg.n_bands;
    ECHO_LOCK(h);
    if (int rc = order_scratch_use(h, s)) return rc;
    for (int f0 = 0; f0 < B; f0 += h->max_chunk) {
        const int n = std::min(h->max_chunk, B - f0);
        int rc = ensure_scratch(h, n);
        if (rc) return rc;
        if (h->long_path) {
            LongParams q = h->lbase;
            q.det_io = corr + (size_t)f0 * nb * q.L; q.det_stride = q.L; q.det_off = q.half; q.write_mutated = 1;
            q.res_out = h->sc.res;
            rc = launch_long_detect(q, n, s);
        } else {
            BandParams p = h->base;
            p.mode = MODE_CORR; p.do_corr = 1; p.write_corr_mutated = 1;
            p.corr_io = corr + (size_t)f0 * nb * p.L; p.res_out = h->sc.res;
            rc = launch_band(p, n, s);
        }
        if (rc) return rc;
        rc = launch_unpack_res(h->sc.res, peak_idx + (size_t)f0 * nb, snr_db + (size_t)f0 * nb,
                               amplitude ? amplitude + (size_t)f0 * nb : nullptr, n * nb, s);
        if (rc) return rc;
    }
    return ECHO_OK;
}

int echo_envelope_detection(EchoHandle* h, const double* corr, double* envelope, int32_t* peak_idx, double* snr_db,
                            double* quality, int32_t* n_peaks, int32_t* peaks, int B, void* stream) {
    if (h && !on_device(h)) return ECHO_ERR_INVALID;         // the handle belongs to another device
    if (h && B == 0) return ECHO_OK;                       // empty batch: nothing to do
    if (!h || !corr || B < 0) return ECHO_ERR_INVALID;
    if (!h->has_env) return ECHO_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nb = h->cfg.n_bands;
    ECHO_LOCK(h);
    if (int rc = order_scratch_use(h, s)) return rc;
    for (int f0 = 0; f0 < B; f0 += h->max_chunk) {
        const int n = std::min(h->max_chunk, B - f0);
        int rc = ensure_scratch(h, n);
        if (rc) return rc;
        EnvParams q = h->ebase;
        const size_t o = (size_t)f0 * nb;
        q.corr = corr + o * q.L; q.env_out = envelope ? envelope + o * q.L : nullptr; q.res_out = h->sc.env;
        q.peaks_out = peaks ? peaks + o * kEnvMaxPeaks : nullptr;
        rc = launch_corr_env(q, n, s);
        if (rc) return rc;
        rc = launch_unpack_env(h->sc.env, peak_idx ? peak_idx + o : nullptr, snr_db ? snr_db + o : nullptr,
                               quality ? quality + o : nullptr, n_peaks ? n_peaks + o : nullptr, n * nb, s);
        if (rc) return rc;
    }
    return ECHO_OK;
}

int echo_argmax(const double* x, int32_t* out, int rows, int n, void* stream) {
    if (!x || !out || rows < 0 || n < 1) return ECHO_ERR_INVALID;
    return launch_argmax(x, out, rows, n, static_cast<hipStream_t>(stream));
}

int echo_confidence_fuse(EchoHandle* h, const int32_t* peak_idx, const double* snr_db, const double* amplitude,
                         EchoResult* out, int B, void* stream) {
    if (h && !on_device(h)) return ECHO_ERR_INVALID;         // the handle belongs to another device
    if (h && B == 0) return ECHO_OK;                       // empty batch: nothing to do
    if (!h || !peak_idx || !snr_db || !amplitude || !out || B < 0) return ECHO_ERR_INVALID;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nb = h->cfg.n_bands;
    ECHO_LOCK(h);
    if (int rc = order_scratch_use(h, s)) return rc;
    for (int f0 = 0; f0 < B; f0 += h->max_chunk) {
        const int n = std::min(h->max_chunk, B - f0);
        int rc = ensure_scratch(h, n);
        if (rc) return rc;
        rc = launch_pack_res(peak_idx + (size_t)f0 * nb, snr_db + (size_t)f0 * nb, amplitude + (size_t)f0 * nb,
                             h->sc.res, n * nb, s);
        if (rc) return rc;
        rc = launch_fuse(fuse_params(h, h->sc.res, out + f0, nullptr, nullptr, n), s);
        if (rc) return rc;
    }
    return ECHO_OK;
}

int echo_kalman_update_qr(double* x, double* p, uint8_t* init, const double* z, const uint8_t* valid, double* out,
                          int S, int T, double q, double r, void* stream) {
    if (!x || !p || !init || !z || !valid || S < 0 || T < 0) return ECHO_ERR_INVALID;
    KalmanParams k;
    k.x = x; k.p = p; k.init = init; k.z = z; k.valid = valid; k.out = out; k.rec = nullptr;
    k.S = S; k.T = T; k.q = q; k.r = r;
    return launch_kalman(k, static_cast<hipStream_t>(stream));
}

int echo_kalman_update(EchoHandle* h, double* x, double* p, uint8_t* init, const double* z, const uint8_t* valid,
                       double* out, int S, int T, void* stream) {
    if (h && !on_device(h)) return ECHO_ERR_INVALID;         // the handle belongs to another device
    if (!h) return ECHO_ERR_INVALID;
    return echo_kalman_update_qr(x, p, init, z, valid, out, S, T, h->cfg.kalman_q, h->cfg.kalman_r, stream);
}

int echo_band_confidence(EchoHandle* h, const double* snr_db, const double* amplitude, const int32_t* band_idx,
                         double* conf, int n, void* stream) {
    if (h && n == 0) return ECHO_OK;
    if (!h || !snr_db || !amplitude || !band_idx || !conf || n < 0) return ECHO_ERR_INVALID;
    if (!on_device(h)) return ECHO_ERR_INVALID;
    return launch_band_confidence(snr_db, amplitude, band_idx, conf, n, h->cfg.confidence, h->cfg.freq_weights,
                                  static_cast<hipStream_t>(stream));
}

int echo_normalize_confidences(const double* conf, double* out, int rows, int n, void* stream) {
    if (rows == 0) return ECHO_OK;
    if (!conf || !out || rows < 0 || n < 1) return ECHO_ERR_INVALID;
    return launch_normalize_conf(conf, out, rows, n, static_cast<hipStream_t>(stream));
}

int echo_xcorr(const double* a, const double* b, double* out, int rows, int n, int m, void* stream) {
    if (rows == 0) return ECHO_OK;
    if (!a || !b || !out || rows < 0 || n < 1 || m < 1) return ECHO_ERR_INVALID;
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int r0 = 0; r0 < rows; r0 += 65535) {
        const int nr = std::min(65535, rows - r0);
        const int rc = launch_xcorr(a + (size_t)r0 * n, b, out + (size_t)r0 * (n + m - 1), nr, n, m, s);
        if (rc) return rc == -2 ? ECHO_ERR_UNSUPPORTED : ECHO_ERR_HIP;
    }
    return ECHO_OK;
}

int echo_spectrum_rows(EchoHandle* h, const double* x, float* spec, int db, int rows, void* stream) {
    if (h && rows == 0) return ECHO_OK;
    if (!h || !x || !spec || rows < 0) return ECHO_ERR_INVALID;
    if (!on_device(h)) return ECHO_ERR_INVALID;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the spectrum kernels take one workgroup per row and no per-band table: rows are passed as frames of one band
    const int step = h->long_path ? 65535 : (1 << 20);
    for (int r0 = 0; r0 < rows; r0 += step) {
        const int nr = std::min(step, rows - r0);
        int rc;
        if (h->long_path) {
            LongParams q = h->lbase;
            q.nb = 1; q.spec_db = db ? 1 : 0;
            q.spec_in = x + (size_t)r0 * q.N; q.spec_out = spec + (size_t)r0 * q.nbins;
            rc = run_long_spec(h, q, nr, s);
        } else {
            BandParams p = h->base;
            p.nb = 1; p.spec_db = db ? 1 : 0;
            p.specd_in = x + (size_t)r0 * p.N; p.spec_out = spec + (size_t)r0 * p.nbins;
            rc = launch_spec(p, nr, s);
        }
        if (rc) return rc;
    }
    return ECHO_OK;
}

int echo_first_strong_peak_rows(double* corr, int rows, int len, int min_delay, int noise_n, double snr_gate_db,
                                int32_t* peak_idx, double* snr_db, void* stream) {
    if (rows == 0) return ECHO_OK;
    if (!corr || !peak_idx || !snr_db || rows < 0 || len < 1 || min_delay < 0 || noise_n < 0) return ECHO_ERR_INVALID;
    const int rc = launch_fsp_rows(corr, rows, len, min_delay, noise_n, snr_gate_db, peak_idx, snr_db,
                                   static_cast<hipStream_t>(stream));
    return rc == 0 ? ECHO_OK : (rc == -2 ? ECHO_ERR_UNSUPPORTED : ECHO_ERR_HIP);
}

// ---- spectra of signals of any length.  A length the FFT kernels hold gets a plan of its own -- a one-band handle with
// the identity filter, of which only the spectrum tables are used -- kept per (device, length); any other length runs
// as direct sums against a table of its n roots of unity.  Plans live until the process ends (at most kSpecPlans of
// them; the oldest goes first).
namespace {
struct SpecPlan { int device = -1, n = 0; EchoHandle* h = nullptr; double2* roots = nullptr; };
constexpr int kSpecPlans = 16;
std::mutex g_spec_mu;
std::vector<SpecPlan> g_spec_plans;

int spec_plan_for(int n, SpecPlan** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return ECHO_ERR_NODEVICE;
    for (auto& pl : g_spec_plans)
        if (pl.device == dev && pl.n == n) { *out = &pl; return ECHO_OK; }
    SpecPlan pl;
    pl.device = dev; pl.n = n;
    if (n >= 64) {
        const double one = 1.0;
        EchoConfig c;
        std::memset(&c, 0, sizeof(c));
        c.fs = 48000; c.n_rx = n; c.n_chirp = 1; c.n_bands = 1; c.iir_ncoef = 1; c.n_taps = 1; c.noise_n = 1;
        c.detector = ECHO_DET_MEDIAN; c.snr_gate_db = 6.0; c.sound_speed = 343.0; c.kalman_q = 0.005; c.kalman_r = 0.1;
        c.b = &one; c.a = &one; c.zi = nullptr; c.taps = &one; c.chirps = &one;
        EchoHandle* h = nullptr;
        if (echo_create(&c, &h) == ECHO_OK) {
            if (h->long_path && !h->lbase.has_spec) { echo_destroy(h); h = nullptr; }   // direct sums need no handle
            pl.h = h;
        }
    }
    if (!pl.h) {
        if (n > kSpecDftMax) return ECHO_ERR_UNSUPPORTED;
        std::vector<double2> w((size_t)n);
        const long double tau = 6.283185307179586476925286766559005768L;
        for (int t = 0; t < n; ++t) {
            const long double a = -tau * (long double)t / (long double)n;
            w[t] = double2{(double)cosl(a), (double)sinl(a)};
        }
        const int rc = dev_upload(&pl.roots, w.data(), w.size());
        if (rc) { (void)hipFree(pl.roots); return rc; }
    }
    if ((int)g_spec_plans.size() >= kSpecPlans) {
        SpecPlan old = g_spec_plans.front();
        g_spec_plans.erase(g_spec_plans.begin());
        if (old.h) echo_destroy(old.h);                     // (hipFree inside waits for work still using the tables)
        (void)hipFree(old.roots);
    }
    g_spec_plans.push_back(pl);
    *out = &g_spec_plans.back();
    return ECHO_OK;
}
}  // namespace

int echo_spectrum_any(const double* x, float* spec, int db, int rows, int n, void* stream) {
    if (rows == 0) return ECHO_OK;
    if (!x || !spec || rows < 0 || n < 1) return ECHO_ERR_INVALID;
    std::lock_guard<std::mutex> guard(g_spec_mu);
    SpecPlan* pl = nullptr;
    int rc = spec_plan_for(n, &pl);
    if (rc) return rc;
    if (pl->h) return echo_spectrum_rows(pl->h, x, spec, db, rows, stream);
    rc = launch_dft_rows(x, pl->roots, spec, db, rows, n, static_cast<hipStream_t>(stream));
    return rc == 0 ? ECHO_OK : ECHO_ERR_HIP;
}

// plan (and device tables) of the envelope kernel for rows of `len` samples smoothed over `smooth`, made once
static int env_plan_for(EchoHandle* h, int len, int smooth, EnvParams** out) {
    for (auto& pl : h->env_plans)
        if (pl.L == len && pl.smooth == smooth) { *out = &pl.p; return ECHO_OK; }
    EchoHandle::EnvPlan e;
    std::memset(&e, 0, sizeof(e));
    e.L = len; e.smooth = smooth;
    if (!plan_env(e.p, len, 1, 0, smooth, h->env_lin != 0)) return ECHO_ERR_UNSUPPORTED;
    std::vector<cpx<double>> er, ewn, ewl, bw, bb;
    int rc = ECHO_OK;
    auto up = [&](int r) { if (rc == ECHO_OK) rc = r; };
    if (e.p.lin) {
        std::vector<double> gam;
        envlin_tables(len, er, ewn, gam);
        up(dev_upload(&e.ltw, er.data(), er.size()));
        up(dev_upload(&e.lwk, ewn.data(), ewn.size()));
        up(dev_upload(&e.lgam, gam.data(), gam.size()));
        std::vector<cpx<double>> wm, gm;
        envlin_mid_tables(ewn, gam, wm, gm);
        up(dev_upload(&e.lwm, wm.data(), wm.size()));
        up(dev_upload(&e.lgm, gm.data(), gm.size()));
    } else {
    env_tables(e.p, er, ewn, ewl);
    up(dev_upload(&e.roots, er.data(), er.size()));
    up(dev_upload(&e.wn, ewn.data(), ewn.size()));
    up(dev_upload(&e.wl, ewl.data(), ewl.size()));
    }
    if (e.p.bs_m) {
        env_bs_tables(e.p, bw, bb);
        up(dev_upload(&e.bsw, bw.data(), bw.size()));
        up(dev_upload(&e.bsb, bb.data(), bb.size()));
    }
    if (rc == ECHO_OK) rc = env_kernel_prepare(e.p);
    if (rc != ECHO_OK) {
        (void)hipFree(e.roots); (void)hipFree(e.wn); (void)hipFree(e.wl); (void)hipFree(e.bsw); (void)hipFree(e.bsb);
        (void)hipFree(e.ltw); (void)hipFree(e.lwk); (void)hipFree(e.lgam); (void)hipFree(e.lwm); (void)hipFree(e.lgm);
        return rc;
    }
    e.p.roots = e.roots; e.p.wn = e.wn; e.p.wl = e.wl; e.p.bs_w = e.bsw; e.p.bs_bhat = e.bsb;
    e.p.lin_tw = e.ltw; e.p.lin_wk = e.lwk; e.p.lin_gam = e.lgam; e.p.lin_wmid = e.lwm; e.p.lin_gmid = e.lgm;
    h->env_plans.push_back(e);
    *out = &h->env_plans.back().p;
    return ECHO_OK;
}

int echo_envelope_rows(EchoHandle* h, const double* x, int len, int smooth_window, int min_delay, double* envelope,
                       int32_t* peak_idx, double* snr_db, double* quality, int32_t* n_peaks, int32_t* peaks, int rows,
                       void* stream) {
    if (h && rows == 0) return ECHO_OK;
    if (!h || !x || rows < 0 || len < 1 || min_delay < 0) return ECHO_ERR_INVALID;
    if (!on_device(h)) return ECHO_ERR_INVALID;
    ECHO_LOCK(h);
    if (h->capturing) return ECHO_ERR_INVALID;
    EnvParams* plan = nullptr;
    int rc = env_plan_for(h, len, smooth_window, &plan);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = order_scratch_use(h, s);
    if (rc) return rc;
    const int step = 65535;
    const size_t need = (size_t)std::min(rows, step);
    if (need > h->env_rows_cap) {
        (void)hipFree(h->d_env_rows); h->d_env_rows = nullptr; h->env_rows_cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&h->d_env_rows), need * sizeof(EnvRes)) != hipSuccess) return ECHO_ERR_NOMEM;
        h->env_rows_cap = need;
    }
    for (int r0 = 0; r0 < rows; r0 += step) {
        const int nr = std::min(step, rows - r0);
        EnvParams q = *plan;
        q.nb = 1; q.min_delay = min_delay;
        q.corr = x + (size_t)r0 * len; q.env_out = envelope ? envelope + (size_t)r0 * len : nullptr;
        q.res_out = h->d_env_rows; q.peaks_out = peaks ? peaks + (size_t)r0 * kEnvMaxPeaks : nullptr;
        rc = launch_corr_env(q, nr, s);
        if (rc) return rc;
        rc = launch_unpack_env(h->d_env_rows, peak_idx ? peak_idx + r0 : nullptr, snr_db ? snr_db + r0 : nullptr,
                               quality ? quality + r0 : nullptr, n_peaks ? n_peaks + r0 : nullptr, nr, s);
        if (rc) return rc;
    }
    return ECHO_OK;
}

// ext: scratch of a single pass provided by the caller of chain_impl (echo_chain_noalloc); null = the handle's own,
// grown on demand.  With ext nothing is allocated, a pending prefetched IIR is neither consumed nor discarded, and no
// event of the handle's is recorded (the call may sit inside a stream capture of the caller's).
static int chain_impl(EchoHandle* h, const float* rx, EchoResult* out, float* spec_db, double* kalman_x,
                      double* kalman_p, uint8_t* kalman_init, int B, int n_streams, void* stream, bool full,
                      double* quality, const Scratch* ext = nullptr, double* corr_out = nullptr,
                      hipEvent_t kal_wait = nullptr, hipEvent_t kal_done = nullptr) {
    if (h && !on_device(h)) return ECHO_ERR_INVALID;         // the handle belongs to another device
    if (h && B == 0) {                                     // empty batch: nothing to do but to pass the Kalman order on
        hipStream_t s0 = static_cast<hipStream_t>(stream);
        if (kal_wait && hipStreamWaitEvent(s0, kal_wait, 0) != hipSuccess) return ECHO_ERR_HIP;
        if (kal_done && hipEventRecord(kal_done, s0) != hipSuccess) return ECHO_ERR_HIP;
        return ECHO_OK;
    }
    if (!h || !rx || !out || B < 0) return ECHO_ERR_INVALID;
    if (full && (h->long_path || !h->has_env || !hilb_fits(h->base.N, h->base.n2))) return ECHO_ERR_UNSUPPORTED;
    const bool kal = kalman_x || kalman_p || kalman_init;
    if (kal && (!kalman_x || !kalman_p || !kalman_init || n_streams < 1 || B % n_streams != 0)) return ECHO_ERR_INVALID;
    if (!detector_supported(h)) return ECHO_ERR_UNSUPPORTED;
    if (h->long_path && spec_db && !long_spec_available(h)) return ECHO_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nb = h->cfg.n_bands, N = h->cfg.n_rx;
    // frames are processed in passes of at most max_chunk; with a Kalman state each pass must hold whole
    // time steps (multiples of n_streams) so the time-major order is preserved
    int chunk = h->max_chunk;
    // Calls of many passes pipeline themselves: passes of 8192 frames, the IIR of pass k+1 on the handle's stream
    // beside the band kernel of pass k.  The first IIR is exposed and the co-running band kernel is slower, so this
    // pays from about eight passes on (2.86 M against 2.76 M frames/s at a million frames).
    const bool inner = !ext && !h->long_path && !h->capturing && h->inner_pipeline && B >= 8 * kInnerChunk &&
                       (!kal || n_streams <= kInnerChunk);
    if (inner) chunk = kInnerChunk;
    if (kal) {
        if (n_streams > chunk) chunk = n_streams;          // one time step per pass at least
        else chunk = (chunk / n_streams) * n_streams;
    }
    if (inner) {
        int rc = ensure_prefetch(h, chunk);
        if (rc) return rc;
        h->pf_valid = false;                               // a pending single-batch prefetch is discarded
        if (hipEventRecord(h->pf_rx_ready, s) != hipSuccess) return ECHO_ERR_HIP;   // rx may be produced on s
        rc = issue_prefetch(h, rx, std::min(chunk, B), h->pf_rx_ready, false);
        if (rc) return rc;
    }
    for (int f0 = 0; f0 < B; f0 += chunk) {
        const int n = std::min(chunk, B - f0);
        int rc = ext ? ECHO_OK : ensure_scratch(h, n);
        if (rc) return rc;
        if (full) { rc = ensure_full_scratch(h, n); if (rc) return rc; }
        const Scratch& sc = ext ? *ext : h->sc;
        const bool prof = !ext && h->prof && f0 == 0 && h->prof_used < h->prof_slots &&
                          (h->prof_calls++ % h->prof_every) == 0;
        hipEvent_t* ev = prof ? &h->ev[kEvPerSlot * h->prof_used] : nullptr;
        bool pair_timed = false;
        if (prof) (void)hipEventRecord(ev[0], s);
        // the IIR result of this batch may already be on its way on the handle's own stream (echo_chain_prefetch)
        const bool pre = inner || (!ext && f0 == 0 && n == B && h->pf_valid && h->pf_rx == rx && h->pf_B == B);
        const double* y_in = sc.y;
        if (!ext && (f0 == 0 || inner)) h->pf_valid = false;   // consumed or discarded (a caller-scratch call leaves it alone)
        const int pre_slot = h->pf_slot;
        bool use_side = false, track = false, fce = false;
        const int side_mode = (full && !h->long_path) ? h->iir_side_full : h->iir_side;
        const int side_cus = (full && !h->long_path) ? h->iir_cus_full : h->iir_cus;
        if (!pre && side_mode && !ext && !h->capturing) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            const bool plain = hipStreamIsCapturing(s, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone;
            track = plain && side_mode == 2;
            // single-pass calls only: a second pass's IIR must wait for the first pass's readers of the scratch, and that
            // wait at the head of the shared stream would hold up the other handles' IIRs queued behind it
            use_side = plain && (side_mode == 1 || (n == B && n >= h->iir_side_min && others_in_flight(h)));
        }
        hipStream_t side = nullptr;
        if (use_side && !(side = side_stream(h->device, side_cus))) {
            // no such stream on this device (a partition mode without compute-unit masks, say): the IIR stays in line
            h->iir_side = h->iir_side_full = 0;
            use_side = track = false;
        }
        if (pre) {
            if (hipStreamWaitEvent(s, h->pf_ready[pre_slot], 0) != hipSuccess) return ECHO_ERR_HIP;
            y_in = h->pf_y[pre_slot];
            if (inner && f0 + chunk < B) {                 // the next pass's IIR goes out before this pass's band kernel
                rc = issue_prefetch(h, rx + (size_t)(f0 + chunk) * N, std::min(chunk, B - f0 - chunk), nullptr, false);
                if (rc) return rc;
                h->pf_valid = false;
            }
        } else if (use_side) {
            if (!h->side_fork && (hipEventCreateWithFlags(&h->side_fork, hipEventDisableTiming) != hipSuccess ||
                                  hipEventCreateWithFlags(&h->side_join, hipEventDisableTiming) != hipSuccess)) return ECHO_ERR_HIP;
            ++h->side_passes;
            std::lock_guard<std::mutex> guard(g_side_mu);
            if (hipEventRecord(h->side_fork, s) != hipSuccess || hipStreamWaitEvent(side, h->side_fork, 0) != hipSuccess) return ECHO_ERR_HIP;
            rc = run_iir(h, rx + (size_t)f0 * N, sc.y, n, side, sc.fwd);
            if (rc) return rc;
            if (hipEventRecord(h->side_join, side) != hipSuccess || hipStreamWaitEvent(s, h->side_join, 0) != hipSuccess) return ECHO_ERR_HIP;
        } else {
            rc = run_iir(h, rx + (size_t)f0 * N, sc.y, n, s, sc.fwd);
            if (rc) return rc;
        }
        if (prof) (void)hipEventRecord(ev[1], s);
        if (h->long_path) {
            LongParams q = h->lbase;
            q.fir_rx = 0; q.d_in = y_in; q.band_out = sc.band;
            rc = launch_long_fir(q, n, s);
            if (rc) return rc;
            q.band_in = sc.band; q.corr_out = sc.pos;
            const bool all_lags = q.detector != ECHO_DET_MAD;           // every other detector reports max|corr| over all lags
            q.k_begin = all_lags ? 0 : std::min(q.half + q.min_delay, q.L - 1);
            q.corr_k0 = all_lags ? 0 : q.half; q.corr_stride = all_lags ? q.L : q.L - q.half;
            q.det_io = sc.pos; q.det_stride = q.corr_stride; q.det_off = 0; q.write_mutated = 0; q.res_out = sc.res;
            if (h->fuse_det && long_rcorr_det_fits(q)) {
                rc = launch_long_rcorr_det(q, n, s);        // one real transform; the correlation stays in the workgroup
            } else if (q.rc_nblk > 1 && long_rcorr_fits(q)) {
                rc = launch_long_rcorr(q, n, s);            // blocks of lags, two workgroups per compute unit; detector apart
                if (rc) return rc;
                rc = launch_long_detect(q, n, s);
            } else if (h->fuse_det && long_corr_det_fits(q)) {
                rc = launch_long_corr_det(q, n, s);         // the correlation stays in the workgroup that detects on it
            } else {
                rc = long_rcorr_fits(q) ? launch_long_rcorr(q, n, s) : launch_long_corr(q, n, s);
                if (rc) return rc;
                rc = launch_long_detect(q, n, s);
            }
            if (rc) return rc;
            if (spec_db) {
                q.spec_db = 1; q.spec_in = sc.band; q.spec_out = spec_db + (size_t)f0 * nb * q.nbins;
                rc = run_long_spec(h, q, n, s);
                if (rc) return rc;
            }
        } else if (full) {
            double* corr = corr_out ? corr_out + (size_t)f0 * nb * h->base.L : h->d_corr;     // raw correlation: the caller's, or scratch
            fce = h->fce_ok && (reinterpret_cast<uintptr_t>(corr) & 15) == 0;
            BandParams p = h->base;
            p.mode = MODE_Y_FIR; p.do_corr = 1; p.corr_fast = 0;
            p.d_in = y_in; p.res_out = sc.res; p.corr_io = corr; p.bandf_out = sc.bandf;
            if (fce) {
                // FIR, correlation (one real cyclic convolution) and the envelope branch in one kernel; first_strong_peak
                // -- direct sums over every lag in float64, the raw correlation rewritten from them -- only for the
                // (frame, band)s whose envelope branch found no peak (airSonar_optimized.py:657)
                EnvParams e = h->ebase;
                // the raw correlation leaves the kernel only for a caller who asked for it
                e.y_in = y_in; e.bandf_out = sc.bandf; e.corr = corr; e.corr_w = corr_out ? corr : nullptr;
                e.env_out = nullptr; e.res_out = sc.env; e.peaks_out = nullptr;
                rc = launch_fce(e, n, s);
                if (rc) return rc;
                p.bandf_out = nullptr;
                p.skip_gate = &sc.env[0].peak_idx; p.skip_stride = (int)(sizeof(EnvRes) / sizeof(int32_t));
            }
            // every lag in float64 (the envelope branch transforms the whole correlation), raw corr kept
            rc = launch_band(p, n, s);
            if (rc) return rc;
        } else {
            BandParams p = h->base;
            p.mode = MODE_Y_FIR; p.do_corr = 1; p.corr_fast = 1;
            p.d_in = y_in; p.res_out = sc.res;
            p.bandf_out = spec_db ? sc.bandf : nullptr;
            p.det_out = sc.det;
            if (!p.band2) rc = launch_band(p, n, s);
            else {
                rc = launch_band2_pair(p, n, s);
                if (rc) return rc;
                if (prof) { (void)hipEventRecord(ev[5], s); pair_timed = true; }
                rc = launch_band2_exact(p, n, s);
            }
            if (rc) return rc;
        }
        if (!ext && h->pf_stream && !h->capturing && h->pf_gate == 2) {
            (void)hipEventRecord(h->band_end, s);
            h->band_end_set = true;
        }
        if (pre) {                                         // the band kernels were the last readers of the prefetched IIR output
            if (hipEventRecord(h->pf_consumed[pre_slot], s) != hipSuccess) return ECHO_ERR_HIP;
            h->pf_consumed_set[pre_slot] = true;
        }
        if (prof) (void)hipEventRecord(ev[2], s);
        if (h->long_path) {
            // spectra of long frames are written by the long-path branch above
        } else if (full) {
            BandParams q = h->base;
            q.spec_db = 1; q.specf_in = sc.bandf; q.sq_out = h->d_sq;
            q.spec_out = spec_db ? spec_db + (size_t)f0 * nb * q.nbins : h->d_spec;
            rc = launch_spec_hilbert(q, n, s);
            if (rc) return rc;
            if (!fce) {
                EnvParams e = h->ebase;
                e.corr = corr_out ? corr_out + (size_t)f0 * nb * e.L : h->d_corr;
                e.env_out = nullptr; e.res_out = sc.env; e.peaks_out = nullptr;
                rc = launch_corr_env(e, n, s);
                if (rc) return rc;
            }
        } else if (spec_db) {
            BandParams q = h->base;
            q.spec_db = 1; q.specf_in = sc.bandf; q.spec_out = spec_db + (size_t)f0 * nb * q.nbins;
            rc = launch_spec(q, n, s);
            if (rc) return rc;
        }
        if (prof) (void)hipEventRecord(ev[3], s);
        FuseParams fp = fuse_params(h, sc.res, out + f0, sc.fused, sc.vany, n);
        if (full) { fp.env = sc.env; fp.sq = h->d_sq; fp.quality_out = quality ? quality + (size_t)f0 * nb * 2 : nullptr; }
        rc = launch_fuse(fp, s);
        if (rc) return rc;
        if (kal) {
            // the Kalman state is the only thing a batch inherits from the one before: a caller running batches on
            // several streams orders this stage alone
            if (kal_wait && f0 == 0 && hipStreamWaitEvent(s, kal_wait, 0) != hipSuccess) return ECHO_ERR_HIP;
            KalmanParams k;
            k.x = kalman_x; k.p = kalman_p; k.init = kalman_init; k.z = sc.fused; k.valid = sc.vany;
            k.out = nullptr; k.rec = out + f0; k.S = n_streams; k.T = n / n_streams;
            k.q = h->cfg.kalman_q; k.r = h->cfg.kalman_r;
            rc = launch_kalman(k, s);
            if (rc) return rc;
        }
        if (track) note_in_flight(h, s);
        if (prof) {
            (void)hipEventRecord(ev[4], s);
            if ((int)h->prof_pair.size() <= h->prof_used) h->prof_pair.resize(h->prof_used + 1);
            h->prof_pair[h->prof_used] = pair_timed;
            ++h->prof_used;
        }
    }
    if (!ext && h->pf_stream && h->pf_gate && !h->capturing) {
        h->chain_gen ^= 1;
        (void)hipEventRecord(h->chain_end[h->chain_gen], s);
        h->chain_end_set[h->chain_gen] = true;
    }
    if (kal_done && hipEventRecord(kal_done, s) != hipSuccess) return ECHO_ERR_HIP;
    return ECHO_OK;
}

// ---- large calls on three streams of the handle's own.
// Every kernel of the chain keeps one unit of the GPU busy at a time, and up to the fusion a frame depends on nothing
// but itself; only the Kalman stage continues what the frames before it left.  A call of at least three chunks
// (4096 frames each: chunks of 8192 were slower at every call size from 12288 to 65536 frames, chunks of 2048 too)
// therefore runs its chunks round-robin on three internal streams --
// lane 0 with the handle's own scratch, lanes 1 and 2 with clones that share the handle's tables -- forked from and
// joined to the caller's stream by events, the Kalman stage of each chunk waiting for the previous chunk's.  The
// records are the bytes of the same call run pass after pass (a frame's results do not depend on its batch).
constexpr int kLaneChunk = 4096, kLaneChunkLong = 4096, kSelfLanes = 3;

static EchoHandle* clone_for_lane(const EchoHandle* h) {
    EchoHandle* c = new (std::nothrow) EchoHandle(*h);
    if (!c) return nullptr;
    c->is_clone = true;
    // nothing the clone could free or reuse of the original's: scratch, prefetch state, events, plans, lanes
    c->sc = Scratch(); c->cap_frames = 0; c->cap_full = 0;
    c->d_corr = nullptr; c->d_sq = nullptr; c->d_spec = nullptr;
    c->pf_stream = nullptr; c->pf_t0 = c->pf_t1 = c->pf_rx_ready = nullptr;
    for (int i = 0; i < 2; ++i) {
        c->pf_ready[i] = c->pf_consumed[i] = c->chain_end[i] = nullptr;
        c->pf_consumed_set[i] = c->chain_end_set[i] = false; c->pf_y[i] = nullptr; c->lane_clone[i] = nullptr;
    }
    c->pf_fwd = nullptr; c->pf_cap = 0; c->pf_rx = nullptr; c->pf_B = 0; c->pf_valid = false;
    c->band_end = nullptr; c->band_end_set = false;
    c->env_plans.clear(); c->d_env_rows = nullptr; c->env_rows_cap = 0;
    c->prof = false; c->ev.clear(); c->prof_pair.clear(); c->prof_slots = c->prof_used = 0;
    c->capturing = false; c->live_graphs = 0; c->self_lanes = 0;
    for (int i = 0; i < 3; ++i) { c->lane_stream[i] = nullptr; c->lane_join[i] = nullptr; c->lane_kal[i] = nullptr; }
    c->lane_fork = nullptr; c->lanes_ready = false;
    c->last_stream = nullptr; c->last_stream_set = false; c->xs_event = nullptr;
    c->side_fork = nullptr; c->side_join = nullptr; c->side_passes = 0; c->counted = false;
    return c;
}

static int ensure_lanes(EchoHandle* h) {
    if (h->lanes_ready) return ECHO_OK;
    // every object is created only if it is still missing: a call that failed half-way is simply continued
    for (int i = 0; i < 2; ++i)
        if (!h->lane_clone[i] && !(h->lane_clone[i] = clone_for_lane(h))) return ECHO_ERR_NOMEM;
    auto event = [](hipEvent_t& e) { return e || hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess; };
    if (!event(h->lane_fork)) return ECHO_ERR_HIP;
    for (int i = 0; i < kSelfLanes; ++i)
        if (!event(h->lane_join[i]) || !event(h->lane_kal[i])) return ECHO_ERR_HIP;
    for (int i = 0; i < kSelfLanes; ++i)
        if (!h->lane_stream[i] && hipStreamCreateWithFlags(&h->lane_stream[i], hipStreamNonBlocking) != hipSuccess)
            return ECHO_ERR_HIP;
    h->lanes_ready = true;
    return ECHO_OK;
}

// chain_impl, or its chunks on the handle's lanes when the call is large enough
static int chain_entry(EchoHandle* h, const float* rx, EchoResult* out, float* spec_db, double* kalman_x, double* kalman_p,
                       uint8_t* kalman_init, int B, int n_streams, void* stream, bool full, double* quality,
                       double* corr_out, hipEvent_t kal_wait, hipEvent_t kal_done) {
    if (!h) return ECHO_ERR_INVALID;
    ECHO_LOCK(h);
    if (on_device(h) && B > 0) {
        if (int rc = order_scratch_use(h, static_cast<hipStream_t>(stream))) return rc;
    }
    const bool kal = kalman_x || kalman_p || kalman_init;
    int chunk = (h && h->long_path) ? kLaneChunkLong : kLaneChunk;
    if (h && kal && n_streams >= 1 && n_streams <= chunk) chunk = (chunk / n_streams) * n_streams;
    const bool lanes = h && on_device(h) && rx && out && h->self_lanes && !h->is_clone && !h->capturing &&
                       h->live_graphs == 0 && B >= kSelfLanes * chunk &&
                       (!kal || (kalman_x && kalman_p && kalman_init && n_streams >= 1 && n_streams <= chunk &&
                                 B % n_streams == 0)) &&
                       !(full && (h->long_path || !h->has_env || !hilb_fits(h->base.N, h->base.n2))) &&
                       detector_supported(h) && !(h->long_path && spec_db && !long_spec_available(h));
    if (!lanes)
        return chain_impl(h, rx, out, spec_db, kalman_x, kalman_p, kalman_init, B, n_streams, stream, full, quality,
                          nullptr, corr_out, kal_wait, kal_done);
    int rc = ensure_lanes(h);
    if (rc) return rc;
    h->pf_valid = false;                                   // a pending single-batch prefetch is discarded
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipEventRecord(h->lane_fork, s) != hipSuccess) return ECHO_ERR_HIP;
    for (int i = 0; i < kSelfLanes; ++i)
        if (hipStreamWaitEvent(h->lane_stream[i], h->lane_fork, 0) != hipSuccess) return ECHO_ERR_HIP;
    const int nb = h->cfg.n_bands, N = h->cfg.n_rx;
    const size_t nbins = h->long_path ? (size_t)h->lbase.nbins : (size_t)h->base.nbins;
    const size_t L = (size_t)h->base.L;
    hipEvent_t prev = kal_wait;
    int i = 0;
    for (int f0 = 0; f0 < B; f0 += chunk, ++i) {
        const int n = std::min(chunk, B - f0), lane = i % kSelfLanes;
        const bool last = f0 + chunk >= B;
        EchoHandle* hl = lane == 0 ? h : h->lane_clone[lane - 1];
        hipEvent_t done = kal ? ((last && kal_done) ? kal_done : h->lane_kal[lane]) : nullptr;
        rc = chain_impl(hl, rx + (size_t)f0 * N, out + f0, spec_db ? spec_db + (size_t)f0 * nb * nbins : nullptr,
                        kalman_x, kalman_p, kalman_init, n, n_streams, h->lane_stream[lane], full,
                        quality ? quality + (size_t)f0 * nb * 2 : nullptr, nullptr,
                        corr_out ? corr_out + (size_t)f0 * nb * L : nullptr, kal ? prev : nullptr, done);
        if (rc) break;
        if (kal) prev = done;
    }
    // The caller's stream continues when every lane is through -- also after an error, so that nothing is left behind:
    // every lane is joined whatever the others' joins return, and the first error of the call is the one reported.
    int join_rc = ECHO_OK;
    for (int k = 0; k < kSelfLanes; ++k) {
        if (hipEventRecord(h->lane_join[k], h->lane_stream[k]) != hipSuccess ||
            hipStreamWaitEvent(s, h->lane_join[k], 0) != hipSuccess) {
            // this lane cannot be ordered before the caller's stream by an event: wait for it here instead
            (void)hipStreamSynchronize(h->lane_stream[k]);
            if (join_rc == ECHO_OK) join_rc = ECHO_ERR_HIP;
        }
    }
    if (rc) return rc;
    if (join_rc) return join_rc;
    if (!kal && kal_done && hipEventRecord(kal_done, s) != hipSuccess) return ECHO_ERR_HIP;
    return ECHO_OK;
}

int echo_chain_ordered(EchoHandle* h, const float* rx, EchoResult* out, float* spec_db, double* kalman_x,
                       double* kalman_p, uint8_t* kalman_init, int B, int n_streams, int full, void* stream,
                       void* kalman_wait_event, void* kalman_done_event) {
    if (!kalman_x || !kalman_p || !kalman_init) return ECHO_ERR_INVALID;      // without a Kalman stage there is nothing to order
    return chain_entry(h, rx, out, spec_db, kalman_x, kalman_p, kalman_init, B, n_streams, stream, full != 0, nullptr,
                       nullptr, static_cast<hipEvent_t>(kalman_wait_event), static_cast<hipEvent_t>(kalman_done_event));
}

int echo_chain(EchoHandle* h, const float* rx, EchoResult* out, float* spec_db, double* kalman_x, double* kalman_p,
               uint8_t* kalman_init, int B, int n_streams, void* stream) {
    return chain_entry(h, rx, out, spec_db, kalman_x, kalman_p, kalman_init, B, n_streams, stream, false, nullptr,
                       nullptr, nullptr, nullptr);
}

int echo_chain_full(EchoHandle* h, const float* rx, EchoResult* out, float* spec_db, double* quality, double* kalman_x,
                    double* kalman_p, uint8_t* kalman_init, int B, int n_streams, void* stream) {
    return chain_entry(h, rx, out, spec_db, kalman_x, kalman_p, kalman_init, B, n_streams, stream, true, quality,
                       nullptr, nullptr, nullptr);
}

int echo_chain_wave(EchoHandle* h, const float* rx, EchoResult* out, float* spec_db, double* corr, double* quality,
                    double* kalman_x, double* kalman_p, uint8_t* kalman_init, int B, int n_streams, void* stream) {
    if (!spec_db || !corr) return ECHO_ERR_INVALID;
    return chain_entry(h, rx, out, spec_db, kalman_x, kalman_p, kalman_init, B, n_streams, stream, true, quality, corr,
                       nullptr, nullptr);
}

int echo_chain_noalloc(EchoHandle* h, const float* rx, EchoResult* out, float* spec_db, double* kalman_x,
                       double* kalman_p, uint8_t* kalman_init, int B, int n_streams, void* stream) {
    if (!h || B < 0) return ECHO_ERR_INVALID;
    if (B == 0) return ECHO_OK;
    ECHO_LOCK(h);
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != h->device || !h->noalloc) return ECHO_ERR_INVALID;
    if (B > h->max_chunk) return ECHO_ERR_TOO_LARGE;        // one internal pass only
    Scratch sc;
    if (!carve_scratch(h, B, h->noalloc, kNoallocBytes, sc)) return ECHO_ERR_TOO_LARGE;
    return chain_impl(h, rx, out, spec_db, kalman_x, kalman_p, kalman_init, B, n_streams, stream, false, nullptr, &sc);
}

int echo_noalloc_max_frames(EchoHandle* h) {
    if (!h) return ECHO_ERR_INVALID;
    int lo = 0, hi = h->max_chunk;                          // largest B whose scratch fits the static array
    Scratch sc;
    while (lo < hi) {
        const int mid = (lo + hi + 1) / 2;
        if (carve_scratch(h, mid, h->noalloc, kNoallocBytes, sc)) lo = mid; else hi = mid - 1;
    }
    return lo;
}

int echo_synth_frames(EchoHandle* h, const float* tx_mix, float* rx, int B, long long first_frame,
                      unsigned long long seed, void* stream) {
    if (h && !on_device(h)) return ECHO_ERR_INVALID;         // the handle belongs to another device
    if (!h || !tx_mix || !rx || B < 0 || first_frame < 0) return ECHO_ERR_INVALID;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int N = h->cfg.n_rx;
    for (int f0 = 0; f0 < B; f0 += 65535) {                 // one workgroup per frame
        const int n = std::min(65535, B - f0);
        const int rc = launch_synth(rx + (size_t)f0 * N, tx_mix, n, N, h->cfg.n_chirp, h->cfg.fs, first_frame + f0, seed,
                                    343.42f, s);
        if (rc) return rc;
    }
    return ECHO_OK;
}

int echo_chain_prefetch(EchoHandle* h, const float* rx, int B) {
    if (h && !on_device(h)) return ECHO_ERR_INVALID;         // the handle belongs to another device
    if (!h || !rx || B < 1) return ECHO_ERR_INVALID;
    if (B > h->max_chunk) return ECHO_ERR_UNSUPPORTED;
    ECHO_LOCK(h);
    int rc = ensure_prefetch(h, B);
    if (rc) return rc;
    return issue_prefetch(h, rx, B, nullptr, true);
}

int echo_prefetch_ms(EchoHandle* h, float* iir_ms) {
    if (!h || !iir_ms) return ECHO_ERR_INVALID;
    ECHO_LOCK(h);
    if (!h->pf_timed) return ECHO_ERR_INVALID;
    if (hipEventSynchronize(h->pf_t1) != hipSuccess || hipEventElapsedTime(iir_ms, h->pf_t0, h->pf_t1) != hipSuccess)
        return ECHO_ERR_HIP;
    return ECHO_OK;
}

struct EchoGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    EchoHandle* owner = nullptr;
};

int echo_graph_create(EchoHandle* h, const float* rx, EchoResult* out, float* spec_db, double* kalman_x,
                      double* kalman_p, uint8_t* kalman_init, int B, int n_streams, void* stream, EchoGraph** graph) {
    if (!h || !graph || !stream || B < 1 || B > h->max_chunk || !on_device(h)) return ECHO_ERR_INVALID;
    *graph = nullptr;
    ECHO_LOCK(h);
    int rc = ensure_scratch(h, B);          // allocate before the capture: nothing inside it may allocate
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool prof = h->prof;
    h->prof = false;                        // timing events are not part of the recorded work
    h->pf_valid = false;                    // nor is a pending prefetch: the graph always runs its own IIR
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) { h->prof = prof; return ECHO_ERR_HIP; }
    h->capturing = true;
    rc = echo_chain(h, rx, out, spec_db, kalman_x, kalman_p, kalman_init, B, n_streams, stream);
    h->capturing = false;
    EchoGraph* g = new (std::nothrow) EchoGraph();
    const hipError_t e = hipStreamEndCapture(s, g ? &g->graph : nullptr);
    h->prof = prof;
    if (!g) return ECHO_ERR_NOMEM;
    if (rc != ECHO_OK || e != hipSuccess || !g->graph) { echo_graph_destroy(g); return rc ? rc : ECHO_ERR_HIP; }
    if (hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0) != hipSuccess) {
        echo_graph_destroy(g);
        return ECHO_ERR_HIP;
    }
    g->owner = h;
    ++h->live_graphs;
    *graph = g;
    return ECHO_OK;
}

int echo_graph_launch(EchoGraph* g, void* stream) {
    if (!g || !g->exec || !g->owner) return ECHO_ERR_INVALID;
    ECHO_LOCK(g->owner);                    // the recorded kernels use the owner's scratch
    if (int rc = order_scratch_use(g->owner, static_cast<hipStream_t>(stream))) return rc;
    return hipGraphLaunch(g->exec, static_cast<hipStream_t>(stream)) == hipSuccess ? ECHO_OK : ECHO_ERR_HIP;
}

void echo_graph_destroy(EchoGraph* g) {
    if (!g) return;
    if (g->owner) { ECHO_LOCK(g->owner); --g->owner->live_graphs; }
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}

}  // extern "C"

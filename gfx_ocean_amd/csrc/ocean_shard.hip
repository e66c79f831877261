// ocean_shard.hip -- C ABI of ONE N x N transform sharded by row blocks over the GPUs of a node
// (include/ocean_hip.h "sharded tile").  The reference has no counterpart: its transform is one
// 512 x 512 dispatch chain on one GPU (src/render.rs:1122-1310); the only ordering it fixes -- all row passes
// before any column pass, barrier at :1181-1208 -- becomes the all-to-all between ocean_shard_rows and
// ocean_shard_cols.  The collective itself is the caller's (torch.distributed / RCCL in gfx_ocean_amd/sharded.py):
// this file only fills the send buffer and consumes the receive buffer.
//
// Included at the end of ocean_api.hip (one translation unit: the kernels of ocean_kernels.hpp are defined once).

struct OceanShard {
    int device = 0, n = 0, rank = 0, world = 1, rows = 0;      // rows = columns per rank = n / world
    hipStream_t stream = nullptr;
    // One event per entry point (rows, cols), recorded behind its most recent launches on whichever stream they ran on:
    // sync / upload / destroy wait for BOTH (rows may run on one caller stream and cols on another; the caller's
    // streams are never named again).
    hipEvent_t last[2] = {nullptr, nullptr};
    bool last_valid[2] = {false, false};
    c32* h0_own = nullptr;        // rows [rank rows, (rank+1) rows) of the initial spectrum
    c32* h0_partner = nullptr;    // rows [n - (rank+1) rows, n - rank rows): where the "-k" partners live (propagate.comp:48)
    float* omega = nullptr;       // own rows of the dispersion
    c32* fld[3] = {nullptr, nullptr, nullptr};    // row pass: own rows of dx, dy, dz; column pass: own columns as lines
    c32* tw = nullptr;
    bool uploaded = false;
    std::string err;
};

namespace {

thread_local std::string g_shard_create_error;
std::mutex g_shard_mu;
std::unordered_set<const void*> g_shard_live;
bool shard_live(const OceanShard* s) { if (!s) return false; std::lock_guard<std::mutex> l(g_shard_mu); return g_shard_live.count(s) != 0; }

int32_t shard_fail(OceanShard* s, int32_t code, const std::string& msg) {
    if (s) s->err = msg; else g_shard_create_error = msg;
    return code;
}
int32_t shard_hip_fail(OceanShard* s, hipError_t e, const char* what) {
    return shard_fail(s, e == hipErrorOutOfMemory ? OCEAN_E_OOM : OCEAN_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define SHARD_TRY(s, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return shard_hip_fail((s), e_, #expr); } while (0)

struct ShardDeviceGuard {
    int prev = -1;
    explicit ShardDeviceGuard(int dev) { (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~ShardDeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

bool shard_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
int shard_log2i(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// line transforms of every supported length (512 .. 16384: one 16384-point line is 1024 threads and 139 KiB of LDS)
template <int N> struct ShardLaunch {
    static constexpr int E = 16, T = N / E;
    static constexpr int LPW = (256 / T) > 1 ? (256 / T) : 1;
    static constexpr int lds = LPW * LinePitch<N>::elems * (int)sizeof(c32);
    static hipError_t prepare() {
        hipError_t e = hipFuncSetAttribute((const void*)k_shard_rows<N, E, LPW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        return hipFuncSetAttribute((const void*)k_fft_lines<N, E, LPW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    static void rows(OceanShard* s, int f, c32* send, hipStream_t st) {
        hipLaunchKernelGGL((k_shard_rows<N, E, LPW>), dim3(s->rows / LPW), dim3(T * LPW), lds, st, (const c32*)s->fld[f], send,
                           (const c32*)s->tw, f, s->rows, shard_log2i(s->rows));
    }
    static void lines(OceanShard* s, int f, hipStream_t st) {
        hipLaunchKernelGGL((k_fft_lines<N, E, LPW, false>), dim3(s->rows / LPW), dim3(T * LPW), lds, st, s->fld[f], (const c32*)s->tw);
    }
};
#define SHARD_DISPATCH(n, STMT)                                    \
    switch (n) {                                                   \
        case 512: { using L = ShardLaunch<512>; STMT; } break;     \
        case 1024: { using L = ShardLaunch<1024>; STMT; } break;   \
        case 2048: { using L = ShardLaunch<2048>; STMT; } break;   \
        case 4096: { using L = ShardLaunch<4096>; STMT; } break;   \
        case 8192: { using L = ShardLaunch<8192>; STMT; } break;   \
        case 16384: { using L = ShardLaunch<16384>; STMT; } break; \
        default: break;                                            \
    }

void shard_free_all(OceanShard* s) {
    auto f = [](void* p) { if (p) (void)hipFree(p); };
    f(s->h0_own); f(s->h0_partner); f(s->omega); f(s->fld[0]); f(s->fld[1]); f(s->fld[2]); f(s->tw);
    for (hipEvent_t e : s->last) if (e) (void)hipEventDestroy(e);
    if (s->stream) (void)hipStreamDestroy(s->stream);
}
// Everything this shard has launched, on its own stream or a caller's, has completed.
hipError_t shard_wait_all(OceanShard* s) {
    for (int i = 0; i < 2; ++i)
        if (s->last_valid[i]) { hipError_t e = hipEventSynchronize(s->last[i]); if (e != hipSuccess) return e; }
    return hipStreamSynchronize(s->stream);
}

}  // namespace

extern "C" {

int32_t ocean_shard_create(int32_t device, int32_t resolution, int32_t rank, int32_t world, OceanShard** out) {
    if (!out) return shard_fail(nullptr, OCEAN_E_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!shard_pow2(resolution) || resolution < 512 || resolution > 16384)
        return shard_fail(nullptr, OCEAN_E_UNSUPPORTED_N, "sharded tile: resolution must be a power of two in [512, 16384]");
    if (!shard_pow2(world) || rank < 0 || rank >= world || resolution / world < 32)
        return shard_fail(nullptr, OCEAN_E_INVALID_ARG, "sharded tile: world must be a power of two with at least 32 rows per rank, 0 <= rank < world");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) return shard_hip_fail(nullptr, e, "hipGetDeviceCount");
    if (device < 0 || device >= count) return shard_fail(nullptr, OCEAN_E_INVALID_ARG, "no such HIP device");
    OceanShard* s = new (std::nothrow) OceanShard();
    if (!s) return shard_fail(nullptr, OCEAN_E_OOM, "host allocation failed");
    s->device = device; s->n = resolution; s->rank = rank; s->world = world; s->rows = resolution / world;
    ShardDeviceGuard guard(device);
    const size_t block = (size_t)s->rows * resolution;
    auto bail = [&](hipError_t err, const char* what) { const int32_t c = shard_hip_fail(nullptr, err, what); shard_free_all(s); delete s; return c; };
#define CREATE_TRY(expr) do { hipError_t e2_ = (expr); if (e2_ != hipSuccess) return bail(e2_, #expr); } while (0)
    CREATE_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreateWithFlags(&s->last[0], hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&s->last[1], hipEventDisableTiming));
    CREATE_TRY(hipMalloc((void**)&s->h0_own, block * sizeof(c32)));
    CREATE_TRY(hipMalloc((void**)&s->h0_partner, block * sizeof(c32)));
    CREATE_TRY(hipMalloc((void**)&s->omega, block * sizeof(float)));
    for (int f = 0; f < 3; ++f) CREATE_TRY(hipMalloc((void**)&s->fld[f], block * sizeof(c32)));
    CREATE_TRY(hipMalloc((void**)&s->tw, (size_t)resolution * sizeof(c32)));
    {
        std::vector<c32> tw((size_t)resolution);
        for (int i = 0; i < resolution; ++i) {
            const double a = 2.0 * M_PI * (double)i / (double)resolution;
            tw[(size_t)i] = mk((float)std::cos(a), (float)std::sin(a));
        }
        CREATE_TRY(hipMemcpy(s->tw, tw.data(), tw.size() * sizeof(c32), hipMemcpyHostToDevice));
    }
    {
        hipError_t pe = hipSuccess;
        SHARD_DISPATCH(resolution, pe = L::prepare());
        if (pe != hipSuccess) return bail(pe, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
#undef CREATE_TRY
    { std::lock_guard<std::mutex> l(g_shard_mu); g_shard_live.insert(s); }
    *out = s;
    return OCEAN_OK;
}

void ocean_shard_destroy(OceanShard* s) {
    if (!shard_live(s)) return;
    { std::lock_guard<std::mutex> l(g_shard_mu); g_shard_live.erase(s); }
    ShardDeviceGuard guard(s->device);
    (void)shard_wait_all(s);
    shard_free_all(s);
    delete s;
}

const char* ocean_shard_last_error(const OceanShard* s) { return shard_live(s) ? s->err.c_str() : g_shard_create_error.c_str(); }
void* ocean_shard_stream(OceanShard* s) { return shard_live(s) ? (void*)s->stream : nullptr; }

int32_t ocean_shard_upload(OceanShard* s, const float* h0_own_rows, const float* h0_partner_rows, const float* omega_own_rows) {
    if (!shard_live(s)) return OCEAN_E_INVALID_ARG;
    if (!h0_own_rows || !h0_partner_rows || !omega_own_rows) return shard_fail(s, OCEAN_E_INVALID_ARG, "NULL input");
    ShardDeviceGuard guard(s->device);
    const size_t block = (size_t)s->rows * s->n;
    SHARD_TRY(s, shard_wait_all(s));              // frames in flight (on any stream) still read the old inputs
    SHARD_TRY(s, hipMemcpy(s->h0_own, h0_own_rows, block * sizeof(c32), hipMemcpyHostToDevice));
    SHARD_TRY(s, hipMemcpy(s->h0_partner, h0_partner_rows, block * sizeof(c32), hipMemcpyHostToDevice));
    SHARD_TRY(s, hipMemcpy(s->omega, omega_own_rows, block * sizeof(float), hipMemcpyHostToDevice));
    s->uploaded = true;
    return OCEAN_OK;
}

int32_t ocean_shard_rows(OceanShard* s, const OceanPropagateLocals* locals, void* send_device, void* stream) {
    if (!shard_live(s)) return OCEAN_E_INVALID_ARG;
    if (!locals || !send_device) return shard_fail(s, OCEAN_E_INVALID_ARG, "NULL argument");
    if (locals->resolution != s->n) return shard_fail(s, OCEAN_E_INVALID_ARG, "PropagateLocals.resolution != tile resolution");
    if (!(locals->domain_size > 0.0f)) return shard_fail(s, OCEAN_E_INVALID_ARG, "domain_size must be > 0");
    if (!s->uploaded) return shard_fail(s, OCEAN_E_STATE, "ocean_shard_upload has not been called");
    ShardDeviceGuard guard(s->device);
    hipStream_t st = stream ? (hipStream_t)stream : s->stream;
    const size_t block = (size_t)s->rows * s->n;
    // shader/propagate.comp:42-72 on the rank's rows (the partner texels come from the opposite row block)
    hipLaunchKernelGGL(k_propagate, dim3((unsigned)((block / 2 + 255) / 256)), dim3(256), 0, st, (const c32*)s->h0_own,
                       (const c32*)s->h0_partner, (const float*)s->omega, s->fld[OCEAN_FIELD_DY], s->fld[OCEAN_FIELD_DX],
                       s->fld[OCEAN_FIELD_DZ], s->n, s->rank * s->rows, s->rows, locals->time, locals->domain_size,
                       (uint32_t)OCEAN_QUIRKS_REFERENCE);
    // shader/fft_row.comp:44-63 on the rank's rows, stored as the all-to-all send buffer
    for (int f = 0; f < 3; ++f) SHARD_DISPATCH(s->n, L::rows(s, f, (c32*)send_device, st));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return shard_hip_fail(s, e, "ocean_shard_rows launch");
    SHARD_TRY(s, hipEventRecord(s->last[0], st));
    s->last_valid[0] = true;
    return OCEAN_OK;
}

int32_t ocean_shard_cols(OceanShard* s, const void* recv_device, void* out_rgba_T_device, void* stream) {
    if (!shard_live(s)) return OCEAN_E_INVALID_ARG;
    if (!recv_device || !out_rgba_T_device) return shard_fail(s, OCEAN_E_INVALID_ARG, "NULL argument");
    if (reinterpret_cast<uintptr_t>(out_rgba_T_device) & 15u) return shard_fail(s, OCEAN_E_INVALID_ARG, "output must be 16-byte aligned");
    if (!s->uploaded) return shard_fail(s, OCEAN_E_STATE, "ocean_shard_upload has not been called");
    ShardDeviceGuard guard(s->device);
    hipStream_t st = stream ? (hipStream_t)stream : s->stream;
    const int cols = s->rows;
    for (int f = 0; f < 3; ++f) {
        // the received column block, row-major, becomes `cols` contiguous lines; shader/fft_col.comp:44-63 on them
        hipLaunchKernelGGL(k_shard_transpose, dim3((unsigned)((s->n / 32) * (cols / 32))), dim3(256), 32 * 33 * sizeof(c32), st,
                           (const c32*)recv_device, s->fld[f], s->n, f, s->rows, cols);
        SHARD_DISPATCH(s->n, L::lines(s, f, st));
    }
    // shader/correction.comp:24-35 on the column block: out[(x - x0) * N + y]
    const size_t block = (size_t)cols * s->n;
    hipLaunchKernelGGL(k_correct, dim3((unsigned)((block / 2 + 255) / 256)), dim3(256), 0, st, (const c32*)s->fld[OCEAN_FIELD_DY],
                       (const c32*)s->fld[OCEAN_FIELD_DX], (const c32*)s->fld[OCEAN_FIELD_DZ], (float4*)out_rgba_T_device, s->n,
                       s->rank * cols, cols);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return shard_hip_fail(s, e, "ocean_shard_cols launch");
    SHARD_TRY(s, hipEventRecord(s->last[1], st));
    s->last_valid[1] = true;
    return OCEAN_OK;
}

int32_t ocean_shard_sync(OceanShard* s) {
    if (!shard_live(s)) return OCEAN_E_INVALID_ARG;
    ShardDeviceGuard guard(s->device);
    SHARD_TRY(s, shard_wait_all(s));
    return OCEAN_OK;
}

}  // extern "C"

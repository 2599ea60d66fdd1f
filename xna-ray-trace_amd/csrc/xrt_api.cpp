// xrt_api.cpp — the C-ABI of include/xrt.h: scene upload, HBM residency, and the per-frame wavefront
// schedule that replaces the body of RayTracer.RenderInternal (RT:105-120).  Compiled by hipcc (host side
// uses the HIP runtime API); all arithmetic of the path runs in kernels.hip.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/xrt.h"
#include "kernels.h"
#include "rccl_gather.h"
#include "scene_host.h"

using namespace xrt;

namespace {

thread_local std::string g_err = "";
int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHECK(expr)                                                                                         \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? XRT_E_OOM : XRT_E_HIP, "%s failed: %s (%s:%d)", #expr,     \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                            \
    } while (0)

// Paths in flight per chunk (multiple of 512 * 16).  ~400 bytes of work buffers per path: 1920x1080 at 16 samples per
// pixel (33.2 M paths) is one chunk of 13 GB per frame context -- sized for 288 GB of HBM, so that whole frames take the
// single-chunk path (no host round trips, two frames overlapping).
constexpr int MAX_CHUNK_PATHS = 1 << 25;
constexpr int HEAP_RAY_CAP = 1 << 22;      // rays per generation of a ray-tree chunk

// roctx ranges around the stages of a frame, for `rocprofv3 --marker-trace --kernel-trace`: XRT_ROCTX=1 loads the
// marker library on first use (no load-time dependency, nothing is called otherwise).  The ranges bracket the ENQUEUE of a
// stage on the host; the kernels they enqueue carry the same names in the kernel trace.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    bool on = false;
    Roctx() {
        if (!getenv("XRT_ROCTX")) return;
        for (const char *n : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            if (void *h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) {
                push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) { on = true; return; }
            }
        }
    }
};
Roctx &roctx() { static Roctx r; return r; }
struct Range {   // RAII: a named range for the enclosing scope
    bool on;
    explicit Range(const char *fmt, int k = 0) : on(roctx().on) {
        if (!on) return;
        char buf[64];
        snprintf(buf, sizeof(buf), fmt, k);
        roctx().push(buf);
    }
    ~Range() { if (on) roctx().pop(); }
};

// No C++ exception may cross the C boundary (a P/Invoke, ctypes or C host would be terminated): the entry points that
// allocate host memory from caller-given sizes run inside this guard.
template <class F>
int guarded(const char *fn, F &&f) {
    try { return f(); }
    catch (const std::bad_alloc &) { return fail(XRT_E_OOM, "%s: out of host memory", fn); }
    catch (const std::exception &e) { return fail(XRT_E_INVALID_ARG, "%s: %s", fn, e.what()); }
    catch (...) { return fail(XRT_E_INTERNAL, "%s: unknown exception", fn); }
}

// XRT_GUARD=1 (read by xrt_scene_create / xrt_scene_load; a test and debugging mode): every device buffer allocated from then on gets
// GUARD_BYTES of a known pattern behind its last element, and the end of every frame and of every batched query checks that the pattern
// is intact -- a kernel that writes past an array it was given is then XRT_E_INTERNAL naming the buffer's size, not a corrupted
// neighbour or a GPU fault somewhere else.  (Round 3 sized the generation-0 arrays by the root box's screen rectangle while one of them
// was still indexed by path: a process abort in the GPU suite that the next edit hid.  tests/test_gpu_parity.py runs the frame modes
// under the guards.)
constexpr size_t GUARD_BYTES = 4096;
constexpr unsigned char GUARD_PATTERN = 0xA5;
std::atomic<int> g_guardMode{0};
struct GuardRegistry {
    std::mutex m;
    std::unordered_map<void *, size_t> bytesOf;   // buffer -> payload bytes (the guard follows)
} g_guards;
inline int guard_alloc(void **p, size_t bytes) {
    const bool on = g_guardMode.load() != 0;
    HIPCHECK(hipMalloc(p, bytes + (on ? GUARD_BYTES : 0)));
    if (on) {
        HIPCHECK(hipMemset((char *)*p + bytes, GUARD_PATTERN, GUARD_BYTES));
        std::lock_guard<std::mutex> lk(g_guards.m);
        g_guards.bytesOf[*p] = bytes;
    }
    return XRT_OK;
}
inline void guard_free(void *p) {
    if (g_guardMode.load() != 0) { std::lock_guard<std::mutex> lk(g_guards.m); g_guards.bytesOf.erase(p); }
    (void)hipFree(p);
}
// All guards of the process (buffers of every scene on the CURRENT device are readable; others are skipped on error).
int guards_check(const char *where) {
    if (g_guardMode.load() == 0) return XRT_OK;
    if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return XRT_OK; }
    std::lock_guard<std::mutex> lk(g_guards.m);
    std::vector<unsigned char> tail(GUARD_BYTES);
    for (const auto &kv : g_guards.bytesOf) {
        if (hipMemcpy(tail.data(), (const char *)kv.first + kv.second, GUARD_BYTES, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); continue; }
        for (size_t i = 0; i < GUARD_BYTES; i++)
            if (tail[i] != GUARD_PATTERN)
                return fail(XRT_E_INTERNAL, "%s: a kernel wrote %zu bytes past the end of a device buffer of %zu bytes (XRT_GUARD)", where, i + 1, kv.second);
    }
    return XRT_OK;
}

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;   // elements
    int ensure(size_t n) {
        if (n <= cap && p) return XRT_OK;
        if (p) { guard_free(p); p = nullptr; cap = 0; }
        if (n == 0) n = 1;
        int rc = guard_alloc((void **)&p, n * sizeof(T));
        if (rc != XRT_OK) { p = nullptr; return rc; }
        cap = n;
        return XRT_OK;
    }
    void release() { if (p) guard_free(p); p = nullptr; cap = 0; }
};

template <class T>
int upload(DevBuf<T> &b, const std::vector<T> &v) {
    int rc = b.ensure(v.size());
    if (rc != XRT_OK) return rc;
    if (!v.empty()) HIPCHECK(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return XRT_OK;
}

}  // namespace

// One host thread per replica device (in-library multi-GPU, xrt_render_opts.n_gpus), created with the replica and parked on a
// condition variable between frames: it has made its device current once and enqueues that device's share of every frame.
// (Round 2 spawned and joined n-1 std::threads per frame: tens of microseconds of host time on a 0.75 ms frame.)
struct RankWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool posted = false, finished = false, quit = false;
    explicit RankWorker(int device) {
        th = std::thread([this, device] {
            (void)hipSetDevice(device);
            std::unique_lock<std::mutex> lk(m);
            for (;;) {
                cv.wait(lk, [this] { return posted || quit; });
                if (quit) return;
                posted = false;
                lk.unlock();
                job();
                lk.lock();
                finished = true;
                cv.notify_all();
            }
        });
    }
    void post(std::function<void()> f) {
        { std::lock_guard<std::mutex> lk(m); job = std::move(f); finished = false; posted = true; }
        cv.notify_all();
    }
    void wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [this] { return finished; }); }
    ~RankWorker() {
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

constexpr int MAX_STAMP_ROWS = 256;   // traversal launches of one frame that can time themselves (kernels.h STAMP_*)
struct xrt_scene {
    int device = -1;   // -1: host-only scene (inspection of the built trees; every compute call fails)
    HostScene hs;
    const HostScene *host = &hs;   // what the frame code reads; a replica on another device points at its primary's
    // HBM-resident scene
    DevBuf<f4> blocks, refN, snodes, shade, leafNB, leafTB, scull, runTB, triTB;
    DevBuf<float> refT, pblocks, lrec;
    DevBuf<g3> refG;
    DevBuf<int> childDfs, srefs, objMesh, runBase;
    DevBuf<MeshRec> meshes;
    DevBuf<ObjRec> objects;
    DevBuf<MaterialRec> materials;
    DevBuf<uint32_t> texels;
    SceneView view{};
    bool resident = false;
    int numCUs = 256;
    int stackNeeded = 2;
    int blocksPerCU = 1, blocksPerCUMesh = 1, blocksPerCUPacket = 1;
    bool packetOk = false;   // the scene's rays can take the wave-packet kernel (one body, one mesh with a real octree)
    // Which ray populations take the wave-packet kernel.  -1 (default): all three of a frame with 16 sub-rays per pixel -- a wave
    // then holds 4 pixels x 16 samples, rays that visit the same leaves (measured on the 1M-triangle frame: 7.3 against 11.1 ms);
    // none otherwise (64 pixels of a 1-sample frame fan out over too many leaves: 5 x slower than the per-lane kernel).
    // XRT_PACKET=<mask> forces it: bit 0 primary rays, 1 shadow rays, 2 closest-hit rays of later generations, 3 seam-1 batches,
    // 4 bits 1 and 2 also apply beyond generation 1 (default: the first two generations and the first shadow rays only).
    int packetMask = -1;
    int packetMaskHeap = -1;   // the same for ray-tree frames (XRT_PACKET_HEAP; -1: by image size)
    // Largest guided batch of k_intersect (XRT_BATCH_MAX).  Round 1 let a wave reserve up to 512 rays per atomic; per-wave clocks
    // (make WAVE_TIMES=1, tools/wave_times.py) showed the median wave of a C3 / C4 launch leaving at 57 % of the launch and the
    // tail growing with the frame size -- waves stuck with eight expensive rays per lane while the queue was empty.  64: C3 3.9 ->
    // 3.4 ms, C4 11.4 -> 9.4 ms per blocking frame; 32 and 16 lose to contention on the queue word.
    int batchMax = 64;
    // Sizes of the last finished single-chunk frame's generations (rays of traversal step k, work items of shade step k): the
    // launches of the next frame of the same geometry are sized for four times that instead of for the whole chip -- a generation
    // of a few thousand rays costs its kernels' launch floor (5-6 us each with full grids, C2: 0.119 -> 0.11 ms).  Sizing only.
    long long genKey = -1, genRays[68], genShade[68];
    int batchMin = 64;         // XRT_BATCH_MIN (development)
    // A launch of fewer rays than 64 per resident wave is dealt evenly over all waves in multiples of spreadMin instead of 64 to a
    // wave: a batch takes as long as its slowest ray and longer the more rays diverge in it, and idle waves cost nothing -- the ten
    // launches of a ray-tree frame of the reference's default scene: 1.05 -> 0.65 ms (XRT_SPREAD_MIN=64: as before).
    int spreadMin = 4;
    int heavyShift = 3;        // listed long rays are dealt one in 2^n work items (0: 64 to a wave); scene_upload: 0 for two-level scenes; XRT_HEAVY_SHIFT
    bool heavyShiftGiven = false;
    bool packetMerge = true;   // the closest-hit and the shadow packets of a step share one launch (XRT_PK_MERGE=0: two launches, as round 2)
    bool noAnswerAtEmission = false;   // XRT_AE=0: k_shade emits every ray (kernels.h ShadeArgs::ae off)
    int packetPrefetch = -1;   // XRT_PK_PREFETCH: -1 launches of fewer than packetPrefetchBelow packets per resident wave prefetch (kernels.h PacketArgs::prefetch), 0 never, 1 always
    int packetPrefetchBelow = 12;
    bool packetBundle = true;  // XRT_PK_BUNDLE=0: no bundle prefilter (kernels.h PacketArgs::bundle)
    int packetCullMin = 4;     // XRT_PK_CULL_MIN (development): leaves with fewer references skip the tight-box test
    // Split walks (packet.hip): one-body scenes; a packet / an item that has walked for this many microseconds looks for pending subtrees to hand to other waves
    // (XRT_PK_SPLIT=0: off; XRT_PK_BUDGET / XRT_PK_BUDGET_ITEM in microseconds; XRT_PK_SPLIT_ITEMS: capacity of a frame context's arena)
    int lvlCheckedTilesX = 0; long long lvlCheckedTiles = 0;   // (LvlMap::inv verified for this frame geometry)
    bool noLevelMap = false;   // XRT_LEVEL_MAP=0: level records for every path of the frame (as before round 4's last build)
    bool packetSplit = false;   // (measured: no gain yet -- profiles/r04/split_walks.txt; XRT_PK_SPLIT=1 switches the split-walk variant of the packet kernel on)
    int packetBudgetUs = 350, packetBudgetItemUs = 150, packetSplitItems = 8192;
    int packetLongUs = 0, packetBudgetLongUs = 8;    // XRT_PK_LONG / XRT_PK_BUDGET_LONG (block entries, whatever the names say): a packet that made more than the first in the context's last frame hands subtrees over every <second> block entries from the start (XRT_PK_LONG=0: no prediction)
    unsigned splitSerial = 0;
    std::map<int, std::pair<DevBuf<unsigned>, DevBuf<unsigned>>> apiSplit;   // seam 1 (testing aid, XRT_PACKET & 8): an arena per stream
    int packetGrabMax = 2;     // XRT_PK_GRAB (development): 8 -> 2 shortened the tail of a launch (C5 blocking 9.0 -> 7.8 ms); 1 loses to contention on the queue word
    int packetStaticDiv = 4;   // XRT_PK_STATIC (development): 1/2 .. 1/8 measured within 2 % of each other on C5
    int sceneMode = MODE_SCENE;   // MODE_SINGLE when the scene is one SceneObject with one Mesh
    hipStream_t stream = nullptr;
    // per-frame work buffers
    DevBuf<xrt_ray> apiRays;
    DevBuf<xrt_hit> apiHits;
    DevBuf<unsigned> queues;
    DevBuf<uint32_t> outRGBA;
    DevBuf<float> outF32;
    DevBuf<unsigned long long> counters;
    // Work buffers of one frame in flight.  Two sets (FrameCtx) so that two frames on two streams can overlap: a launch
    // of persistent waves leaves the machine half empty while its last rays finish, and the other frame's launches fill it.
    struct WorkBufs {
        DevBuf<xrt_ray> rays0, rays1, shadowRays;
        DevBuf<xrt_hit> hits, shadowHits;
        DevBuf<int> path0, path1, index0, heavyList, cnts;
        DevBuf<int> node0, node1, heapFlag;     // ray-tree frames: heap node of every ray; heapFlag[0]: a generation overflowed its buffers
        DevBuf<float> ref0, ref1, lvlAlpha;     // ... refraction index of the medium a ray travels in; alpha per level record
        DevBuf<int> hitFlags0, shadowFlags;   // hit / miss word per ray of hits, shadowHits (a miss has no record)
        DevBuf<unsigned> splitCost;               // ... what every packet of every packet launch of the context's last plain frame cost (PacketArgs::splitCost)
        size_t splitCostStride = 0;               // (packets a launch may have; a frame of another size starts the memory afresh)
        DevBuf<unsigned> splitItems, splitRecs;   // split walks (kernels.h PacketArgs::splitItems): the arena of this context's packet launches (they run one after the other)
        DevBuf<int> shadowOut;                // ShadeArgs::ae: where the answer of the i-th emitted shadow ray goes (slot * lights + light)
        DevBuf<int> shadowFlags1;             // ShadeArgs::ae: part A of step k answers some shadow queries of generation k ITSELF while part B of the same launch still reads
                                              // generation k-1's words: the generations alternate between shadowFlags and this
        DevBuf<unsigned long long> stamps;      // device-clock stamps of the traversal launches (device_util.h), STAMP_STRIDE per launch
        DevBuf<SlotRec> slot0, slot1;
        DevBuf<int> slotNode0, slotNode1;   // ray-tree frames: the node of a slot's hit
        DevBuf<f4> lvlA, lvlB;
        DevBuf<uint32_t> sampleColor;
        DevBuf<float> sampleF32;
        DevBuf<LightRec> lights;
        // adaptive supersampling in flight (RT:170-311 without host round trips): the quadrant levels' buffers belong to the frame context
        struct Level { DevBuf<uint32_t> color; DevBuf<int> childBase, childMask; DevBuf<float> cx, cy; } levels[8];
        bool levelWordsClean = false;           // the level-count words at the head of cnts are zero (k_resolve cleared them)
        bool heapFlagClean = false;
        bool cntsClean = false;                 // cnts is all zero (the previous frame's epilogue cleared what it counted)
        std::vector<LightRec> lightsOnDevice;   // what `lights` holds
        const void *lightsDevPtr = nullptr;
        hipStream_t stream = nullptr;           // the context's own stream (used when the caller passes none)
        hipStream_t lastStream = nullptr;       // the stream the context's last frame ran on
        void release() {
            rays0.release(); rays1.release(); shadowRays.release(); hits.release(); shadowHits.release();
            path0.release(); path1.release(); index0.release(); heavyList.release(); cnts.release(); stamps.release(); hitFlags0.release(); shadowFlags.release(); shadowOut.release(); shadowFlags1.release();
            node0.release(); node1.release(); heapFlag.release(); ref0.release(); ref1.release(); lvlAlpha.release(); slot0.release(); slot1.release(); slotNode0.release(); slotNode1.release();
            lvlA.release(); lvlB.release(); sampleColor.release(); sampleF32.release(); lights.release(); splitItems.release(); splitRecs.release(); splitCost.release(); splitCostStride = 0;
            for (auto &l : levels) { l.color.release(); l.childBase.release(); l.childMask.release(); l.cx.release(); l.cy.release(); }
            if (stream) (void)hipStreamDestroy(stream);
            stream = nullptr;
        }
    };
    // cost feedback (kernels.hip long_ray): per path and generation, what the ray cost in the last frames
    DevBuf<unsigned> costMap;
    size_t costMapPaths = 0;
    unsigned epoch = 100;
    int costT[66];            // per generation: rays that cost more than this are started first; steered in frame_finish
    int longFracLo = 2, longFracHi = 6;   // percent of a generation's rays the list is steered to (XRT_LONG_FRAC=lo,hi)
    bool deepMeshes = false;  // some mesh has a real octree: rays can be long
    float heavyPath = 0.0f;   // rays longer than this inside the root box are traced first (0: off); XRT_HEAVY=<fraction of the box diagonal>
    std::string waveTimesPath;
    std::string stampDumpPath;   // XRT_STAMP_DUMP=<file>: the stamp rows of the last frame (start, waves, every wave's end) -- how long a launch's waves lived
    DevBuf<unsigned long long> waveTimes;   // XRT_WAVE_TIMES=<file>: per-wave clocks of the last frame's launches (development aid)
    // Per-frame host state.  Two contexts so that the next frame can be enqueued while the previous one's counters
    // and timings are still on their way back (xrt_render_device_begin / _end).
    struct FrameCtx {
        std::vector<hipEvent_t> events;
        void *pinned = nullptr;      // host staging for the counter read-back
        size_t pinnedBytes = 0;
        std::vector<std::pair<size_t, size_t>> pairs;   // (start, stop) event indices of the k_intersect launches
        size_t ev = 0;
        hipEvent_t done = nullptr;   // recorded after the frame's last copy
        std::vector<LightRec> hostLights;
        bool pending = false;
        bool fast = false;           // no copy / fill / event-record commands: k_compose hands the counters over, events ride on kernels
        int *pinnedDev = nullptr;    // device view of `pinned`
        long long framePaths = 0;    // paths of the frame (part) this context holds: key of the grid hints
        int frameW = 0, frameH = 0;  // the frame's size in pixels
        bool heap = false, redone = false;   // a ray-tree frame; ... that overflowed on the optimistic way and was rendered again
        bool adaptiveFast = false;           // an adaptive frame enqueued without host round trips (level sizes stay on the device)
        int cntBase = 0, levelCap = 0, quality = 0;   // words in front of the per-pass counters in `pinned`; quadrant capacity of a deeper level
        xrt_camera redoCam; xrt_render_opts redoOpts; std::vector<xrt_light> redoLights;
        uint32_t *redoOut = nullptr; float *redoOutF32 = nullptr; hipStream_t redoSt = nullptr;
        int stampRows = 0;           // traversal launches of the frame that timed themselves (device_util.h)
        unsigned long long *stampHost = nullptr, *stampHostDev = nullptr;   // their (start, end) clock pairs: mapped pinned memory and its device view
        // deferred accounting
        int tallyChunks = 0, cntStride = 0, R = 0, nL = 0;
        bool ae = false;             // ShadeArgs::ae: rays answered at emission are not in the ray lists
        unsigned long long answered = 0;   // ... their number (frame_finish)
        bool collect = false;
        unsigned long long shaded = 0, closestDeep = 0, livePaths = 0, live0 = 0, validPixels = 0;
        size_t liveCap = 0;   // room of the generation-0 ray arrays (k_raygen writes no live ray past it: a count above it is a wrong bound, reported)
        unsigned long long hcnt[2 * C_COUNT] = {0};
        WorkBufs w;
    } frames[8];   // context of ticket `slot`, part j of its frame: frames[slot + 2 * j] (a frame may be split into up to four bands on as many streams)
    std::vector<hipEvent_t> events;   // xrt_scene_intersect timing
    int firstBatch = 64;
    long long heapRayCap = HEAP_RAY_CAP;         // XRT_HEAP_RAY_CAP=<n> forces small ray buffers (tests of the overflow / retry path)
    long long shadowBytes = 8LL << 30;   // budget of a frame context's shadow rays / hits / words (XRT_SHADOW_BYTES): many lights shrink the chunk
    long long maxChunkPaths = MAX_CHUNK_PATHS;   // XRT_CHUNK_PATHS=<n> (multiple of 8192) forces smaller chunks (tests of the multi-chunk path)
    float lastFrameMs = 0.0f;    // GPU time of the last finished frame
    float overlapMinMs = 0.05f;  // frames at least this long run on per-context streams
    // A launch of persistent waves leaves the machine half empty while its last rays finish; a blocking single frame (what the
    // C# host's RenderInternal asks for) has no other frame to fill the gaps, so it is rendered as two halves of its tiles on
    // two streams.  XRT_SPLIT=0 never, 1 frames nobody else overlaps (default), 2 also pipelined frames.  It paid while a launch's
    // waves were alive 55-60 % of its duration (C4 13.3 -> 10.9 ms), did not in rounds 2 and 3 (the second set of launches cost what the
    // overlap gained: C3 2.74 vs 2.91 ms, C4 7.1 vs 6.9, C5 7.9 vs 8.0), and pays again now that the kernels are faster and a launch's tail
    // is a larger share of it (round 4, one box: C3 1.95 -> 1.72 ms per blocking frame, C4 4.26 -> 4.13, C5 4.48 -> 4.43; three or four
    // bands no better; profiles/r04/frame_split.txt).  By default only two-level scenes: the two extra frame contexts cost a one-body scene like C5
    // 6 GB of work buffers for 1 %.
    int splitMode = 1, splitParts = 2;
    bool splitGiven = false;     // XRT_SPLIT was set: else only two-level scenes are split (a one-body scene gains 1-2 % for two more frame contexts' work buffers)
    bool launchEvents = false;   // XRT_LAUNCH_EVENTS=1: single-chunk frames time their traversal launches with events on the dispatch packets, too
    int maxStampRows = MAX_STAMP_ROWS;   // XRT_STAMP_ROWS=<n> (tests): launches of a frame beyond the n-th carry events instead
    bool adaptiveFastOk = true;  // adaptive frames are enqueued whole (level buffers sized optimistically) until a level overflows; XRT_ADAPTIVE_FAST=0
    long long adaptiveCap = 0;   // XRT_ADAPTIVE_CAP=<quadrants> (tests): capacity of the deeper levels instead of one quadrant per pixel
    bool heapFastOk = true;      // single-chunk ray-tree frames go the optimistic way (no host round trip) until one overflows; XRT_HEAP_FAST=0
    bool noGridHints = false;    // XRT_GRID_HINTS=0: every launch is sized for the whole chip
    bool noLaunchTiming = false; // XRT_LAUNCH_TIMING=0: single-chunk frames do not time their traversal launches (xrt_stats.ms_intersect = 0)
    int wallClockKHz = 0;        // rate of the device clock the launches stamp (hipDeviceAttributeWallClockRate)
    float splitMinMs = 1.0f;
    int tune[4] = {24, 16, 48, 32};   // refill threshold (idle lanes), octree-child steps and leaf steps per outer iteration
    bool tuneGiven = false;           // XRT_TUNE was set: keep it
    std::atomic<bool> busy{false};
    std::atomic<float> progress{0.0f};
    // Seam 1 (xrt_scene_intersect / xrt_mesh_intersect / xrt_generate_primary_rays) is re-entrant like the reference's
    // ISpatialManager.GetRayIntersection (ISM:15, called from N render threads, RT:105-113): the host-buffer calls share
    // one staging area and are serialised by this mutex; every stream has its own work-queue word.
    // In-library multi-GPU (xrt_render_opts.n_gpus): copies of the scene on devices device+1 .. (owned), the RCCL
    // communicators, and per ticket the buffer the tile shards are gathered into.  XRT_FAKE_GPUS=1 (test boxes with one
    // GPU): the replicas live on the scene's own device and the exchange is RCCL send-to-self.
    std::vector<xrt_scene *> replicas;
    std::vector<std::unique_ptr<RankWorker>> workers;   // workers[i - 1] drives replica i
    int visibleDevices = 0;                             // hipGetDeviceCount at xrt_scene_create
    RcclGather rccl;
    bool fakeGpus = false;
    DevBuf<uint32_t> gathered[2];    // primary: n * tiles_per_rank * 512 pixels, rank-major
    DevBuf<uint32_t> tileOut[2];     // replica: its tiles of the frame in slot 0 / 1
    DevBuf<uint32_t> frameOut[2];    // W*H frame of a host-output ticket
    hipEvent_t tilesReady[2] = {nullptr, nullptr};   // replica (fake mode): its tiles are rendered
    hipEvent_t tailDone[2] = {nullptr, nullptr};     // primary: gather + de-tile + host copy of the ticket are done
    // Cost-aware tile assignment (xrt.h xrt_scene_set_tile_table / xrt_scene_tile_costs).  tileTable: the installed table (host copy and device
    // copy) for frames of tableW x tableH pixels with tableCount shards, tableTpr slots per rank; tileCost: ticks per LOCAL tile slot of the
    // frames rendered since the last reset, with the geometry they were rendered under (costKey) and their slots' tiles (costTiles).
    std::vector<int> tileTable;
    DevBuf<int> tileTableDev;
    int tableW = 0, tableH = 0, tableCount = 0, tableTpr = 0;
    DevBuf<unsigned> tileCost;
    std::vector<int> costTiles;      // tile of every local slot the cost words belong to
    int costW = 0, costH = 0;
    std::vector<float> balanceCost;  // n_gpus > 1 with balance_tiles: the last frame's costs by tile (all ranks summed), its size
    int balanceW = 0, balanceH = 0, balanceN = 0;
    struct OpenFrame { int nGpus = 0, nParts = 1; bool tail = false, balance = false; uint32_t *hostOut = nullptr, *devOut = nullptr; size_t px = 0; hipStream_t st0 = nullptr; } open[2];
    std::mutex apiMutex;
    std::unordered_map<hipStream_t, int> queueOfStream;
    // development switches, read once at xrt_scene_create (never per frame)
    bool noRectCull = false, oneStream = false, noFeedback = false;

    ~xrt_scene() {
        workers.clear();   // (joins the threads)
        for (xrt_scene *r : replicas) delete r;
        replicas.clear();
        if (device >= 0) {
            (void)hipSetDevice(device);
            tileTableDev.release(); tileCost.release();
            for (int i = 0; i < 2; i++) { if (tilesReady[i]) (void)hipEventDestroy(tilesReady[i]); if (tailDone[i]) (void)hipEventDestroy(tailDone[i]); gathered[i].release(); tileOut[i].release(); frameOut[i].release(); }
            for (auto e : events) (void)hipEventDestroy(e);
            for (auto &f : frames) {
                for (auto e : f.events) (void)hipEventDestroy(e);
                if (f.done) (void)hipEventDestroy(f.done);
                if (f.pinned) (void)hipHostFree(f.pinned);
                if (f.stampHost) (void)hipHostFree(f.stampHost);
                f.w.release();
            }
            if (stream) (void)hipStreamDestroy(stream);
            blocks.release(); leafNB.release(); leafTB.release(); refT.release(); refN.release(); refG.release(); snodes.release(); shade.release();
            childDfs.release(); srefs.release(); scull.release(); runTB.release(); triTB.release(); runBase.release(); pblocks.release(); lrec.release(); objMesh.release(); meshes.release();
            objects.release(); materials.release(); texels.release();
            apiRays.release(); apiHits.release();

            queues.release(); outRGBA.release(); outF32.release(); counters.release(); costMap.release(); waveTimes.release();
        }
    }
};

namespace {

struct BusyGuard {
    xrt_scene *s;
    bool owned;
    explicit BusyGuard(xrt_scene *sc) : s(sc) {
        bool expected = false;
        owned = s->busy.compare_exchange_strong(expected, true);
    }
    ~BusyGuard() { if (owned) s->busy.store(false); }
};

bool in_flight(const xrt_scene *s) {
    if (s->busy.load()) return true;
    for (const auto &f : s->frames) if (f.pending) return true;
    return false;
}

int need_device(xrt_scene *s, const char *fn) {
    if (!s) return fail(XRT_E_INVALID_ARG, "%s: null scene", fn);
    if (s->device < 0) return fail(XRT_E_NO_DEVICE, "%s: host-only scene (created with device -1); libxrt has no CPU execution path", fn);
    if (!s->host->built || !s->resident) return fail(XRT_E_NOT_BUILT, "%s: call xrt_scene_build first", fn);
    HIPCHECK(hipSetDevice(s->device));
    return XRT_OK;
}

hipEvent_t get_event(std::vector<hipEvent_t> &pool, size_t i) {
    while (pool.size() <= i) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        pool.push_back(e);
    }
    return pool[i];
}
hipEvent_t get_event(xrt_scene *s, size_t i) { return get_event(s->events, i); }

int persistent_grid(xrt_scene *s, long long nHost, int raysPerBlock = 256) {
    int full = s->numCUs * s->blocksPerCU;
    if (nHost >= 0) {
        long long want = (nHost + raysPerBlock - 1) / raysPerBlock;   // one block covers >= 256 rays (16 for a guessed size: small launches are spread thin)
        if (want < 1) want = 1;
        if (want < full) return (int)want;
    }
    return full;
}

void fill_stats(xrt_stats *st, const unsigned long long *c /* 2*C_COUNT: closest, shadow */, unsigned long long shaded, unsigned long long pixels) {
    const unsigned long long *a = c, *b = c + C_COUNT;
    st->rays_closest = a[C_RAYS]; st->rays_shadow = b[C_RAYS];
    st->hits_closest = a[C_HITS]; st->hits_shadow = b[C_HITS];
    st->scene_node_tests = a[C_SCENE_NODES] + b[C_SCENE_NODES];
    st->instance_visits = a[C_INSTANCES] + b[C_INSTANCES];
    st->mesh_aabb_tests = a[C_MESH_AABB] + b[C_MESH_AABB];
    st->mesh_queries = a[C_MESH_QUERIES] + b[C_MESH_QUERIES];
    st->mesh_queries_facing_away = a[C_MESH_AWAY] + b[C_MESH_AWAY];
    st->node_tests = a[C_NODES] + b[C_NODES];
    st->leaf_refs = a[C_REFS] + b[C_REFS];
    st->tri_tests = a[C_TRIS] + b[C_TRIS];
    st->shaded_hits = shaded;
    st->pixels = pixels;
    // per query 32 (ray in) + 48 (hit out) + 32 N_node + 4 N_ref + 48 N_tri, two-level terms
    // 32 N_scene_node + 64 N_instance + 24 N_mesh_aabb + 64 per hit (World); shading 76 + 4 per shaded hit; 4 per pixel.
    unsigned long long rays = st->rays_closest + st->rays_shadow, hits = st->hits_closest + st->hits_shadow;
    st->algorithmic_bytes = rays * 80ull + 32ull * st->node_tests + 4ull * st->leaf_refs + 48ull * st->tri_tests +
                            32ull * st->scene_node_tests + 64ull * st->instance_visits + 24ull * st->mesh_aabb_tests + 64ull * hits +
                            80ull * st->shaded_hits + 4ull * pixels;
}

LightRec make_light(const xrt_light &l) {
    LightRec r;
    std::memset(&r, 0, sizeof(r));
    r.kind = l.kind;
    r.px = l.position[0]; r.py = l.position[1]; r.pz = l.position[2];
    r.dx = l.direction[0]; r.dy = l.direction[1]; r.dz = l.direction[2];
    r.cr = l.color[0]; r.cg = l.color[1]; r.cb = l.color[2];
    r.intensity = l.intensity;
    r.angleCosine = (float)std::cos((double)(l.spot_angle * 0.5f));                        // SPOT:25
    r.decayDenom = std::pow((double)(1 - r.angleCosine), (double)l.decay_exponent);        // SPOT:54, constant per light
    if (l.kind == XRT_LIGHT_DIRECTIONAL) { r.px = r.py = r.pz = 0.0f; }                     // DIR:14
    return r;
}

// ---- cost-aware tile assignment (xrt.h) ------------------------------------------------------------------------------------------
// Longest-processing-time-first: tiles in descending order of cost (ties: ascending tile number), each to the rank with the least cost so
// far that still has a free slot (ties: the lowest rank); a rank's slots are filled in that order.  Costs that are not
// positive finite numbers count as the smallest positive cost seen (a tile nobody measured still takes a slot's worth of launch overhead).
int balance_tiles_impl(int tiles, int count, const float *cost, int tpr, int *out) {
    if (tiles <= 0 || count <= 0 || tpr <= 0 || !out) return fail(XRT_E_INVALID_ARG, "xrt_balance_tiles: bad argument");
    if ((long long)tpr * count < tiles) return fail(XRT_E_INVALID_ARG, "xrt_balance_tiles: %d slots per rank x %d ranks do not hold %d tiles", tpr, count, tiles);
    std::fill(out, out + (size_t)count * tpr, -1);
    double lowest = 0.0;
    bool any = false;
    if (cost)
        for (int t = 0; t < tiles; t++)
            if (cost[t] > 0.0f && cost[t] < 3.0e38f) { if (!any || cost[t] < lowest) lowest = cost[t]; any = true; }
    if (!any) {   // no measurement: the round-robin layout of xrt_shard_layout
        for (int t = 0; t < tiles; t++) out[(size_t)(t % count) * tpr + t / count] = t;
        return XRT_OK;
    }
    std::vector<int> order((size_t)tiles);
    std::vector<double> c((size_t)tiles);
    for (int t = 0; t < tiles; t++) { order[(size_t)t] = t; c[(size_t)t] = (cost[t] > 0.0f && cost[t] < 3.0e38f) ? (double)cost[t] : lowest; }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b2) { return c[(size_t)a] > c[(size_t)b2]; });
    std::vector<double> load((size_t)count, 0.0);
    std::vector<std::vector<int>> mine((size_t)count);
    for (int t : order) {
        int best = -1;
        for (int r = 0; r < count; r++)
            if ((int)mine[(size_t)r].size() < tpr && (best < 0 || load[(size_t)r] < load[(size_t)best])) best = r;
        mine[(size_t)best].push_back(t);
        load[(size_t)best] += c[(size_t)t];
    }
    // A rank's slots in the order its tiles were dealt -- descending cost: its launches start with their longest packets.  (A launch ends when
    // its slowest packet does, and a packet of rays that skim the terrain near the horizon takes a hundred times the median: in ascending
    // tile order such packets started wherever their rows fell, and the shards of a frame that were unlucky took up to twice as long as the
    // others whatever their cost sums.)
    for (int r = 0; r < count; r++)
        for (size_t k = 0; k < mine[(size_t)r].size(); k++) out[(size_t)r * tpr + k] = mine[(size_t)r][k];
    return XRT_OK;
}

// Every tile exactly once?
int check_tile_table(int tiles, int count, int tpr, const int *table) {
    if (tpr <= 0 || (long long)tpr * count < tiles) return fail(XRT_E_INVALID_ARG, "tile table: %d slots per rank x %d ranks do not hold %d tiles", tpr, count, tiles);
    std::vector<char> seen((size_t)tiles, 0);
    for (size_t i = 0; i < (size_t)count * tpr; i++) {
        const int t = table[i];
        if (t == -1) continue;
        if (t < 0 || t >= tiles) return fail(XRT_E_INVALID_ARG, "tile table: entry %zu names tile %d of %d", i, t, tiles);
        if (seen[(size_t)t]) return fail(XRT_E_INVALID_ARG, "tile table: tile %d appears twice", t);
        seen[(size_t)t] = 1;
    }
    for (int t = 0; t < tiles; t++) if (!seen[(size_t)t]) return fail(XRT_E_INVALID_ARG, "tile table: tile %d is missing", t);
    return XRT_OK;
}

int make_raygen(const xrt_camera *cam, const xrt_render_opts *o, RayGenParams &g, const float *rootBox = nullptr /* min xyz, max xyz; null: no screen-rectangle cull */) {
    if (cam->vp_width <= 0 || cam->vp_height <= 0) return fail(XRT_E_INVALID_ARG, "viewport must be positive");
    float wv[16], wvp[16], ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    mat_multiply(ident, cam->view, wv);     // Matrix.Multiply(world = Identity, view)   (Viewport.Unproject, RT:415)
    mat_multiply(wv, cam->proj, wvp);       // Matrix.Multiply(.., projection)
    mat_invert(wvp, g.m);                   // Matrix.Invert
    g.vpX = (float)cam->vp_x; g.vpY = (float)cam->vp_y; g.vpW = (float)cam->vp_width; g.vpH = (float)cam->vp_height;
    g.minDepth = cam->vp_min_depth; g.depthRange = cam->vp_max_depth - cam->vp_min_depth;
    g.width = cam->vp_width; g.height = cam->vp_height;
    g.tilesX = (g.width + XRT_TILE_W - 1) / XRT_TILE_W; g.tilesY = (g.height + XRT_TILE_H - 1) / XRT_TILE_H;
    g.shardCount = o->shard_count > 1 ? o->shard_count : 1;
    g.shardRank = o->shard_count > 1 ? o->shard_rank : 0;
    g.samples = (o->use_multisampling == XRT_MS_FIXED16) ? 16 : 1;
    g.quadLevel = -1; g.quadCx = nullptr; g.quadCy = nullptr; g.quadSize = 1.0f;
    g.cullX0 = 0; g.cullY0 = 0; g.cullX1 = g.width - 1; g.cullY1 = g.height - 1;
    g.cullSkipsRecord = 0;
    g.pathsDev = nullptr; g.pathsMul = 1; g.pathsCap = 0;
    g.tileOfSlot = nullptr;
    if (g.shardRank < 0 || g.shardRank >= g.shardCount) return fail(XRT_E_INVALID_ARG, "shard_rank out of range");
    if (rootBox) {
        // Screen rectangle of the scene's root box.  A ray through pixel (x, y) that reaches the box at a point P has P
        // projecting onto (x, y); the box is convex, so with all eight corners in front of the eye every such pixel lies
        // inside the corners' bounding rectangle.  Evaluated in double from the same float matrices; used only when
        // every corner has a clearly positive w and comes back through the inverse matrix k_raygen uses (its pixel and
        // depth unprojected again) to within a thousandth of the box diagonal.
        bool ok = true;
        double x0 = 1e300, y0 = 1e300, x1 = -1e300, y1 = -1e300;
        for (int c = 0; c < 8 && ok; c++) {
            const double p[3] = {rootBox[(c & 1) ? 3 : 0], rootBox[(c & 2) ? 4 : 1], rootBox[(c & 4) ? 5 : 2]};
            double v[4];
            for (int j = 0; j < 4; j++) v[j] = p[0] * wvp[j] + p[1] * wvp[4 + j] + p[2] * wvp[8 + j] + wvp[12 + j];
            const double scale = std::fabs(v[0]) + std::fabs(v[1]) + std::fabs(v[2]) + std::fabs(v[3]);
            if (!(v[3] > 1e-3 * scale) || !(scale < 1e30)) { ok = false; break; }
            const double sx = (v[0] / v[3] + 1.0) * 0.5 * g.vpW + g.vpX, sy = (1.0 - v[1] / v[3]) * 0.5 * g.vpH + g.vpY;
            {   // round trip through g.m
                const double n[4] = {v[0] / v[3], v[1] / v[3], v[2] / v[3], 1.0};
                double u[4];
                for (int j = 0; j < 4; j++) u[j] = n[0] * g.m[j] + n[1] * g.m[4 + j] + n[2] * g.m[8 + j] + g.m[12 + j];
                const double ddx = rootBox[3] - (double)rootBox[0], ddy = rootBox[4] - (double)rootBox[1], ddz = rootBox[5] - (double)rootBox[2];
                const double tol = 1e-3 * std::sqrt(ddx * ddx + ddy * ddy + ddz * ddz);
                if (!(std::fabs(u[3]) > 1e-30) || !(std::fabs(u[0] / u[3] - p[0]) <= tol && std::fabs(u[1] / u[3] - p[1]) <= tol && std::fabs(u[2] / u[3] - p[2]) <= tol)) { ok = false; break; }
            }
            x0 = std::min(x0, sx); x1 = std::max(x1, sx); y0 = std::min(y0, sy); y1 = std::max(y1, sy);
        }
        if (ok && x0 <= x1 && y0 <= y1) {
            const double mx = 2.0 + 1e-3 * g.vpW, my = 2.0 + 1e-3 * g.vpH;
            // (k_raygen's pixel coordinates are the screen coordinates Viewport.Unproject is given, RT:415)
            const double fx0 = std::floor(x0 - mx), fy0 = std::floor(y0 - my), fx1 = std::ceil(x1 + mx), fy1 = std::ceil(y1 + my);
            g.cullX0 = (int)std::max(0.0, std::min(fx0, (double)g.width)); g.cullY0 = (int)std::max(0.0, std::min(fy0, (double)g.height));
            g.cullX1 = (int)std::min((double)g.width - 1, std::max(fx1, -1.0)); g.cullY1 = (int)std::min((double)g.height - 1, std::max(fy1, -1.0));
        }
    }
    return XRT_OK;
}

// The frame: for every chunk of paths  raygen -> [intersect(closest k + shadow k-1) -> shade(A: k, B: k-1)] x (R+2) -> compose,
// then resolve (pixel grid, fixed 16 sub-rays) or the quadrant levels of adaptive supersampling (RT:170-311).
int frame_finish(xrt_scene *s, xrt_scene::FrameCtx &F, xrt_stats *stats);
int split_arena(xrt_scene *s, DevBuf<unsigned> &items, DevBuf<unsigned> &recs, PacketArgs &PA, hipStream_t st);

// Enqueues one frame on `st`.  On return the frame's kernels and its counter read-back are in flight (F.pending);
// frame_finish waits for them.  Adaptive supersampling and ray-tree frames need host decisions between their passes and
// are complete when this returns.
int frame_begin(xrt_scene *s, xrt_scene::FrameCtx &F, const xrt_camera *cam, const xrt_light *lights, int nLights, const xrt_render_opts *opts,
                uint32_t *d_out, float *d_outF32, hipStream_t st, int part = 0, int nParts = 1, bool heapFastAllowed = true) {
    const bool stats = true;   // the read-back is two small pinned copies; always taken
    xrt_scene::WorkBufs &W = F.w;
    if (!cam || !opts || (!lights && nLights > 0) || nLights < 0) return fail(XRT_E_INVALID_ARG, "xrt_render: null argument");
    if (opts->max_reflections < 0 || opts->max_reflections > 64) return fail(XRT_E_INVALID_ARG, "max_reflections out of range");
    if (opts->address_mode < XRT_ADDRESS_CLAMP || opts->address_mode > XRT_ADDRESS_MIRROR)
        return fail(XRT_E_INVALID_ARG, "Value does not fall within the expected range: addressMode (MAT:85)");
    if (opts->filtering != XRT_FILTER_POINT && opts->filtering != XRT_FILTER_BILINEAR)
        return fail(XRT_E_INVALID_ARG, "Value does not fall within the expected range: filtering (MAT:97)");
    const bool heap = s->host->arrays.anyTransparent && opts->max_reflections > 0;   // RT:586-702: binary ray tree
    if (heap && opts->max_reflections > 12)
        return fail(XRT_E_UNSUPPORTED, "Transparent materials with MaxReflections > 12 (a ray tree of more than 8191 nodes per pixel)");
    const int msMode = opts->use_multisampling;
    if (msMode != XRT_MS_OFF && msMode != XRT_MS_FIXED16 && msMode != XRT_MS_ADAPTIVE) return fail(XRT_E_INVALID_ARG, "use_multisampling");
    const bool adaptive = msMode == XRT_MS_ADAPTIVE;
    const int quality = adaptive ? opts->multisample_quality : 0;
    if (adaptive && (quality < 0 || quality > 6)) return fail(XRT_E_UNSUPPORTED, "MultisampleQuality above 6 (4^7 sub-quadrants per pixel)");
    RayGenParams g;
    float rootBox[6] = {0, 0, 0, 0, 0, 0};
    const bool haveRoot = s->host->arrays.snodes.size() >= 2;
    if (haveRoot) {
        const f4 lo = s->host->arrays.snodes[0], hi = s->host->arrays.snodes[1];
        rootBox[0] = lo.x; rootBox[1] = lo.y; rootBox[2] = lo.z; rootBox[3] = hi.x; rootBox[4] = hi.y; rootBox[5] = hi.z;
    }
    int rc = make_raygen(cam, opts, g, (haveRoot && !s->noRectCull) ? rootBox : nullptr);
    if (rc != XRT_OK) return rc;
    if (adaptive) g.samples = 4;
    g.cullSkipsRecord = heap ? 0 : 1;   // (k_compose_tree reads every root record)
    const int R = opts->max_reflections;
    const int nL = nLights;
    const long long totalTiles = (long long)g.tilesX * g.tilesY;
    // an installed tile table for this frame geometry replaces the round-robin layout (xrt.h xrt_scene_set_tile_table)
    const bool tabled = !s->tileTable.empty() && s->tableW == g.width && s->tableH == g.height && s->tableCount == g.shardCount;
    const int *const tableRow = tabled ? s->tileTable.data() + (size_t)g.shardRank * s->tableTpr : nullptr;
    g.tileOfSlot = tabled ? s->tileTableDev.p + (size_t)g.shardRank * s->tableTpr : nullptr;
    const long long myTiles = tabled ? s->tableTpr : shard_tiles_per_rank(totalTiles, g.shardCount, g.tilesX);   // tiles_per_rank (slots, some may be past the end)
    auto tile_of_slot = [&](long long sl) -> long long { return tabled ? (long long)tableRow[sl] : shard_tile(sl, g.shardRank, g.shardCount, g.tilesX); };
    const long long totalPixels = myTiles * 512;
    if (adaptive && totalPixels * 4 > 0x7fffffffLL) return fail(XRT_E_UNSUPPORTED, "frame too large for adaptive supersampling");
    const long long framePaths = totalPixels * g.samples;   // the whole frame (this shard)
    // part `part` of `nParts`: a contiguous range of the frame's paths (whole tiles), rendered by its own frame context
    const long long partStride = nParts > 1 ? ((framePaths / nParts + 8191) / 8192) * 8192 : framePaths;
    const long long partStart = (long long)part * partStride;
    const long long firstPaths = nParts > 1 ? std::max(0LL, std::min(partStride, framePaths - partStart)) : framePaths;
    if (nParts > 1 && (adaptive || firstPaths <= 0)) return fail(XRT_E_INTERNAL, "frame parts need a plain frame");
    // Chunking.  Without refraction a path owns one ray per generation.  With Transparent materials it may own up to
    // 2^k in generation k, but few paths do: chunks are sized optimistically (ray buffers of HEAP_RAY_CAP rays,
    // level records bounded by 8 GB) and a chunk whose generation overflows is retried with a quarter of the paths.
    const size_t nodes = heap ? (((size_t)1 << (R + 1)) - 1) : (size_t)(R + 1);
    long long maxPaths = s->maxChunkPaths;
    // The reference iterates a List<ILight> of any length (RT:534-542).  The shadow rays of one generation are counted in an int
    // (at most 2^30): many lights shrink the chunk, they are not refused -- until not even 8192 paths fit a generation.
    const long long lightBound = (1LL << 30) / (nL > 0 ? nL : 1);
    if (maxPaths > lightBound) maxPaths = lightBound & ~8191LL;
    // ... and in bytes: shadow rays, hits and hit / miss words are 84 bytes per path and light in each frame context.  Many lights shrink
    // the chunk to what XRT_SHADOW_BYTES (default 8 GB per context) holds -- the frame then takes the multi-chunk path -- instead of
    // failing with XRT_E_OOM (a 1080p frame with 100 lights would want 17 GB).
    if (nL > 0) {
        const long long byBytes = ((long long)s->shadowBytes / (84LL * nL)) & ~8191LL;
        if (byBytes < maxPaths) maxPaths = byBytes < 8192 ? 8192 : byBytes;
    }
    if (maxPaths < 8192) return fail(XRT_E_UNSUPPORTED, "%d lights: the shadow rays of 8192 paths do not fit one generation (2^30 rays)", nL);
    if (heap) {
        if (maxPaths > (1 << 21)) maxPaths = 1 << 21;   // (a 1080p frame is one chunk when the level records fit 8 GB: MaxReflections <= 6)
        const long long byRecords = (long long)((8ull << 30) / (nodes * 36ull));
        if (byRecords < maxPaths) maxPaths = byRecords;
        maxPaths &= ~63LL;
        if (maxPaths < 64) maxPaths = 64;
    }
    const long long chunkPaths = firstPaths < maxPaths ? firstPaths : maxPaths;
    const int P = (int)chunkPaths;
    size_t rayCap = (size_t)P;
    F.liveCap = (size_t)P;
    // The per-ray arrays hold LIVE rays only (k_raygen compacts them): a live primary ray belongs to a pixel inside the screen rectangle
    // of the scene's root box, so a plain or 16-sub-ray frame of one chunk needs room for the rectangle's paths, not for every path
    // (C5: 52 % of the image); later generations have fewer rays than the one before.  (Adaptive frames: the deeper quadrant levels are
    // lists of up to a quadrant per pixel; ray trees: sized below.)
    if (!heap && !adaptive && firstPaths <= chunkPaths) {
        const long long rectPaths = (long long)std::max(0, g.cullX1 - g.cullX0 + 1) * (long long)std::max(0, g.cullY1 - g.cullY0 + 1) * g.samples + 64;
        if (rectPaths < (long long)rayCap) { rayCap = (size_t)rectPaths; F.liveCap = rayCap; }
    }
    // ... and the two per-level records (16 bytes each per path and level: the largest arrays of a frame) exist only for the TILES that overlap the rectangle (xrt_core.h
    // LvlMap): plain unsharded frames of one chunk and one part, whose tiles are numbered row by row.  (C5: 3.2 -> 1.7 GB per frame context.)
    size_t lvlStride = (size_t)P;
    std::memset(&g.lvl, 0, sizeof(g.lvl));
    if (!heap && !adaptive && firstPaths <= chunkPaths && nParts == 1 && partStart == 0 && g.shardCount == 1 && !tabled && g.cullSkipsRecord && !s->noLevelMap &&
        g.cullX1 >= g.cullX0 && g.cullY1 >= g.cullY0) {
        const int shift = 9 + (g.samples == 16 ? 4 : (g.samples == 4 ? 2 : 0));
        const int tx0 = g.cullX0 / XRT_TILE_W, tx1 = g.cullX1 / XRT_TILE_W, ty0 = g.cullY0 / XRT_TILE_H, ty1 = g.cullY1 / XRT_TILE_H;
        const long long kept = (long long)(tx1 - tx0 + 1) * (ty1 - ty0 + 1);
        if (kept < totalTiles && (kept << shift) < (long long)P) {
            g.lvl.on = 1; g.lvl.tilesX = g.tilesX; g.lvl.tx0 = tx0; g.lvl.ty0 = ty0; g.lvl.rw = tx1 - tx0 + 1; g.lvl.shift = shift;
            g.lvl.inv = (unsigned)(((1ULL << 32) + (unsigned)g.tilesX - 1) / (unsigned)g.tilesX);
            bool exact = totalTiles < (1LL << 24);
            if (exact && !(s->lvlCheckedTilesX == g.tilesX && s->lvlCheckedTiles >= totalTiles)) {   // (checked once per frame geometry)
                for (long long t = 0; exact && t < totalTiles; t++) exact = (unsigned)(((unsigned long long)t * g.lvl.inv) >> 32) == (unsigned)(t / g.tilesX);
                if (exact) { s->lvlCheckedTilesX = g.tilesX; s->lvlCheckedTiles = totalTiles; }
            }
            if (exact) lvlStride = (size_t)(kept << shift);
            else std::memset(&g.lvl, 0, sizeof(g.lvl));
        }
    }
    if (heap) {
        const size_t capL = (size_t)std::min<long long>(s->heapRayCap, lightBound);   // (>= 8192 >= ... see above; P <= lightBound as well)
        rayCap = (R < 20 && ((size_t)P << R) < capL) ? ((size_t)P << R) : capL;
        if (rayCap < (size_t)P) rayCap = (size_t)P;
    }
    if ((unsigned long long)rayCap * (unsigned long long)(nL > 0 ? nL : 1) > (1ull << 30)) return fail(XRT_E_UNSUPPORTED, "frame too large: %zu rays x %d lights per generation", rayCap, nL);
    const size_t shadowCap = rayCap;   // hits of one generation (each emits nL shadow rays)
    const bool wantF32 = d_outF32 != nullptr && !adaptive && g.samples == 1;
    const bool fuseResolve = !adaptive && !heap && g.samples == 1;   // k_compose writes the framebuffer itself
    // buffers
    if ((rc = W.rays0.ensure(rayCap)) || (rc = W.rays1.ensure(rayCap)) || (rc = W.hits.ensure(rayCap)) || (rc = W.path0.ensure(rayCap)) ||
        (rc = W.path1.ensure(rayCap)) || (rc = W.hitFlags0.ensure(rayCap)) ||
        (rc = W.shadowFlags.ensure(rayCap * (nL > 0 ? nL : 1))) || (rc = W.slot0.ensure(rayCap)) ||
        (rc = W.slot1.ensure(rayCap)) || (rc = W.index0.ensure(rayCap)) || (rc = W.heavyList.ensure(rayCap)) || (rc = W.shadowRays.ensure(rayCap * (nL > 0 ? nL : 1))) ||
        (rc = W.shadowHits.ensure(rayCap * (nL > 0 ? nL : 1))) || (rc = W.lvlA.ensure(lvlStride * nodes)) ||
        (rc = W.lvlB.ensure(lvlStride * nodes)) || (rc = W.sampleColor.ensure(P)) || (rc = W.lights.ensure(nL > 0 ? nL : 1)) ||
        (rc = s->counters.ensure(2 * C_COUNT + 8)))
        return rc;
    if (!s->waveTimesPath.empty() && !s->waveTimes.p) {
        if ((rc = s->waveTimes.ensure((size_t)16 * 3 * 8192))) return rc;
        HIPCHECK(hipMemset(s->waveTimes.p, 0, (size_t)16 * 3 * 8192 * sizeof(unsigned long long)));
    }
    if (heap && ((rc = W.node0.ensure(rayCap)) || (rc = W.node1.ensure(rayCap)) || (rc = W.ref0.ensure(rayCap)) || (rc = W.ref1.ensure(rayCap)) ||
                 (rc = W.slotNode0.ensure(rayCap)) || (rc = W.slotNode1.ensure(rayCap)) ||
                 (rc = W.lvlAlpha.ensure((size_t)P * nodes))))
        return rc;
    if (wantF32 && (rc = W.sampleF32.ensure((size_t)P * 3))) return rc;
    const int cntStride = 4 * (R + 2);          // per chunk: cnt[R+2], scnt[R+2], the long-ray list lengths [R+2], the shadow rays really emitted [R+2] (ShadeArgs::ae)
    constexpr int QW = 1 + 2 * PACKET_QUEUE_WORDS;   // per launch step k: the lane kernel's queue word and the heads of each packet launch (closest, shadow)
    const int qStride = QW * (R + 2);
    // The common frame (one chunk, no supersampling levels, no ray tree, no counting pass) puts nothing but its kernels
    // on the stream: counters come back through k_compose's epilogue and the frame's events ride on raygen / compose.
    // ... and so does a ray-tree frame of one chunk: its buffers are sized optimistically, so the frame carries a word that says
    // "a generation did not fit"; frame_finish then renders it again the careful way (chunks, checks, retries) and the scene's later
    // ray-tree frames take that way from the start.  Not where something is enqueued behind the frame that a redo cannot recall
    // (the in-library gather of n_gpus > 1).
    const bool heapFast = heap && heapFastAllowed && s->heapFastOk && nParts == 1;
    // ... and an adaptive frame (RT:170-311) whose passes are one chunk each: the sizes of the deeper quadrant levels stay on the device
    // (k_ms_decide counts, the next level's kernels read the count), the level buffers are sized optimistically -- one quadrant per
    // pixel and level -- and a level that does not fit sets the same kind of word.
    const bool adaptiveFast = adaptive && !heap && heapFastAllowed && s->adaptiveFastOk && nParts == 1 && !opts->collect_stats && totalPixels * 4 <= chunkPaths &&
                              (quality + 2) * (R + 2) * 2 + 2 <= MAX_STAMP_ROWS;
    const bool fast = (adaptive ? adaptiveFast : (!heap || heapFast)) && firstPaths <= chunkPaths && !opts->collect_stats;
    // Answered at emission (kernels.h ShadeArgs::ae): plain one-chunk frames of one-body scenes whose mesh CAN face away from a ray as a whole (its normal
    // box does not hold the origin).  Not with the counting pass -- it counts the reference's work for every query from the ray lists --, not for ray trees.
    bool ae = fast && !heap && !adaptive && nParts == 1 && s->sceneMode == MODE_SINGLE && s->view.nodeCull != 0 && !s->noAnswerAtEmission;
    if (ae) {
        const MeshRec &m0 = s->host->arrays.meshes[0];
        ae = m0.nbMin[3] == 0.0f && (m0.nbMin[0] > 0.0f || m0.nbMax[0] < 0.0f || m0.nbMin[1] > 0.0f || m0.nbMax[1] < 0.0f || m0.nbMin[2] > 0.0f || m0.nbMax[2] < 0.0f);
    }
    F.ae = ae;
    if (ae && ((rc = W.shadowOut.ensure(rayCap * (nL > 0 ? nL : 1))) || (rc = W.shadowFlags1.ensure(rayCap * (nL > 0 ? nL : 1))))) return rc;
    F.fast = fast;
    F.heap = heap;
    F.redone = false;
    F.adaptiveFast = adaptiveFast && fast;
    F.cntBase = 0;
    if ((heapFast || adaptiveFast) && fast) {   // what a redo needs
        F.redoCam = *cam; F.redoOpts = *opts; F.redoLights.assign(lights, lights + nLights);
        F.redoOut = d_out; F.redoOutF32 = d_outF32; F.redoSt = st;
    }
    F.framePaths = (nParts == 1 && !adaptive) ? firstPaths : -1;
    F.frameW = g.width; F.frameH = g.height;
    // the traversal launches time themselves on the device clock instead of carrying events (device_util.h); a frame of more
    // than MAX_STAMP_ROWS launches (many chunks or supersampling levels) goes on with events
    const bool useStamps = !s->launchEvents && !s->noLaunchTiming && s->wallClockKHz > 0;
    F.stampRows = 0;
    if (useStamps) {
        if ((rc = W.stamps.ensure((size_t)MAX_STAMP_ROWS * STAMP_STRIDE))) return rc;
        if (!F.stampHost) {
            HIPCHECK(hipHostMalloc((void **)&F.stampHost, (size_t)MAX_STAMP_ROWS * 2 * sizeof(unsigned long long), hipHostMallocMapped));
            HIPCHECK(hipHostGetDevicePointer((void **)&F.stampHostDev, F.stampHost, 0));
        }
    }
    if (!st) {
        // No stream given.  Single-chunk frames get a stream per context, so that two frames in flight overlap on the GPU
        // (C2: 0.139 -> 0.115 ms per frame, C3 2.7 -> 2.3); everything else, and frames of a few microseconds, stay on the
        // scene's one stream.  (While every traversal launch carried two events, enqueueing on a stream that had gone idle cost
        // ~15 us a launch and alternating streams made the host the bottleneck of 0.2 ms frames; the launches time themselves
        // now, device_util.h.)
        if (fast && s->lastFrameMs >= s->overlapMinMs && !s->oneStream) {
            if (!W.stream) HIPCHECK(hipStreamCreateWithFlags(&W.stream, hipStreamNonBlocking));
            st = W.stream;
        } else st = s->stream;
    }
    // Tile costs (xrt.h xrt_scene_tile_costs): the packets of plain one-chunk frames add their device-clock ticks to the tile of their first
    // ray.  The words belong to the scene (both frame contexts add to them); a frame of another geometry or tile table starts them afresh.
    unsigned *tileCostDev = nullptr;
    if (fast && !adaptive && !heap && s->packetOk && framePaths < (1LL << 31)) {
        if ((rc = s->tileCost.ensure((size_t)myTiles))) return rc;
        bool same = s->costW == g.width && s->costH == g.height && (long long)s->costTiles.size() == myTiles;
        for (long long sl = 0; same && sl < myTiles; sl++) { const long long t = tile_of_slot(sl); same = s->costTiles[(size_t)sl] == (t < totalTiles ? (int)t : -1); }
        if (!same) {
            s->costTiles.resize((size_t)myTiles);
            for (long long sl = 0; sl < myTiles; sl++) { const long long t = tile_of_slot(sl); s->costTiles[(size_t)sl] = t < totalTiles ? (int)t : -1; }
            s->costW = g.width; s->costH = g.height;
            HIPCHECK(hipMemsetAsync(s->tileCost.p, 0, (size_t)myTiles * sizeof(unsigned), st));
            if (s->frames[0].pending || s->frames[1].pending) HIPCHECK(hipStreamSynchronize(st));   // (the other context's frame may be adding to them: start clean)
        }
        tileCostDev = s->tileCost.p;
    }
    // which ray populations the wave-packet kernel traces this frame (packet.hip): bit 0 primary rays, 1 shadow rays, 2 closest-hit
    // rays of later generations
    // (two-level scenes: at any sample count and in every generation -- what a packet shares there is the scene walk, the bodies'
    // records and the mesh walk of a body, and an 8x8-pixel block of a 1-sample frame sees one or two bodies: C3 2.28 -> 1.57 ms,
    // C4 6.64 -> 4.37 ms per pipelined frame (packets for the first two generations only: 1.78 / 5.0, profiles/r03/packet_masks.txt);
    // one-body scenes: only with 16 sub-rays, and only the first two generations, see above)
    const int pkAuto = s->sceneMode == MODE_SCENE ? 23 : (g.samples >= 16 ? 7 : 0);
    // (ray-tree frames -- Transparent materials -- of two-level scenes: the first two generations and the first shadow rays, where the image
    // is big enough for a launch to be more than its floor: G2 at 720p 1.12 -> 0.97 ms, the reference's default scene at 512x512 0.31 ->
    // 0.40 with packets, so not there; packets in every generation of a ray tree are 1.5-3 x slower (profiles/r03/packet_masks_ray_trees.txt))
    const int pkHeap = s->packetMaskHeap >= 0 ? s->packetMaskHeap : ((s->sceneMode == MODE_SCENE && (long long)g.width * g.height * g.samples >= 600000LL) ? 7 : 0);
    const int pkMask = !s->packetOk ? 0 : (heap ? pkHeap : (s->packetMask >= 0 ? s->packetMask : pkAuto));
    const bool laneClosest = (pkMask & 5) != 5;   // some closest-hit generation is traced ray by ray: the long-ray feedback has a reader
    const bool wantFeedback = fast && !heap && !adaptive && s->deepMeshes && !s->noFeedback && laneClosest;   // (the paths of a deeper quadrant level are a list: no stable key)
    {   // The other context's frame may still be running on another stream.  Two single-chunk frames share nothing they
        // write except scheduling hints; anything else (counting pass, supersampling levels, ray tree, a cost map
        // about to be reallocated or released) runs alone.
        const bool remap = wantFeedback ? (s->costMapPaths != (size_t)framePaths || s->costMap.cap < (size_t)(R + 1) * (size_t)framePaths) : (s->costMap.p != nullptr);
        for (xrt_scene::FrameCtx &O : s->frames)
            if (&O != &F && O.pending && O.w.lastStream != st && (!fast || !O.fast || remap)) HIPCHECK(hipEventSynchronize(O.fast ? O.events[1] : O.done));
        W.lastStream = st;
    }
    if (wantFeedback) {
        const size_t need = (size_t)(R + 1) * (size_t)framePaths;
        if (s->costMapPaths != (size_t)framePaths || s->costMap.cap < need) {   // new frame geometry: forget
            if ((rc = s->costMap.ensure(need))) return rc;
            HIPCHECK(hipMemsetAsync(s->costMap.p, 0, s->costMap.cap * sizeof(unsigned), st));
            s->costMapPaths = (size_t)framePaths;
        }
        if (part == 0) s->epoch++;
    } else if (s->costMap.p) {
        s->costMap.release(); s->costMapPaths = 0;
    }
    if (!fast) {
        W.cntsClean = false;
        HIPCHECK(hipMemsetAsync(s->counters.p, 0, (2 * C_COUNT + 8) * sizeof(unsigned long long), st));
    }
    std::vector<LightRec> &hl = F.hostLights;   // must outlive the asynchronous upload
    hl.assign(nL > 0 ? nL : 1, LightRec());
    for (int i = 0; i < nL; i++) {
        if (lights[i].kind != XRT_LIGHT_SPOT && lights[i].kind != XRT_LIGHT_DIRECTIONAL) return fail(XRT_E_INVALID_ARG, "unknown light kind");
        hl[i] = make_light(lights[i]);
    }
    const bool sameLights = W.lightsOnDevice.size() == (size_t)nL && W.lightsDevPtr == W.lights.p &&
                            (nL == 0 || std::memcmp(W.lightsOnDevice.data(), hl.data(), nL * sizeof(LightRec)) == 0);
    if (nL > 0 && !sameLights) {
        HIPCHECK(hipMemcpyAsync(W.lights.p, hl.data(), nL * sizeof(LightRec), hipMemcpyHostToDevice, st));
        W.lightsOnDevice.assign(hl.begin(), hl.begin() + nL);
        W.lightsDevPtr = W.lights.p;
    }
    ShadeView V;
    V.shade = s->shade.p; V.materials = s->materials.p; V.texels = s->texels.p; V.meshes = s->meshes.p;
    V.lights = W.lights.p; V.nLights = nL; V.addressMode = opts->address_mode; V.filtering = opts->filtering;
    const SceneView &S = s->view;
    xrt_ray *rays[2] = {W.rays0.p, W.rays1.p};
    int *paths[2] = {W.path0.p, W.path1.p};
    int *nodesOf[2] = {W.node0.p, W.node1.p};
    float *refOf[2] = {W.ref0.p, W.ref1.p};
    size_t &ev = F.ev;
    ev = 0;
    std::vector<std::pair<size_t, size_t>> &pairs = F.pairs;
    pairs.clear();
    hipEvent_t e0 = get_event(F.events, ev++), e1 = get_event(F.events, ev++);
    if (!F.done && hipEventCreateWithFlags(&F.done, hipEventDisableTiming) != hipSuccess) F.done = nullptr;
    if (!e0 || !e1 || !F.done) return fail(XRT_E_HIP, "hipEventCreate failed");
    if (!fast) HIPCHECK(hipEventRecord(e0, st));
    s->progress.store(0.0f);
    unsigned long long &shaded = F.shaded, &closestDeep = F.closestDeep, &livePaths = F.livePaths, &live0 = F.live0;
    shaded = closestDeep = livePaths = live0 = 0;
    F.answered = 0;
    unsigned long long *hcntHost = F.hcnt;
    std::memset(hcntHost, 0, sizeof(F.hcnt));
    F.tallyChunks = 0; F.cntStride = cntStride; F.R = R; F.nL = nL; F.collect = opts->collect_stats != 0;

    // One pass = trace `total` paths produced by generator `gp`; after every chunk `post(Pc, pathBase)` consumes sampleColor.
    // "a generation did not fit its buffers": ray-tree frames have the word to themselves (two of them may be in flight)
    if (heap) {
        const bool fresh = W.heapFlag.p == nullptr;
        if ((rc = W.heapFlag.ensure(16))) return rc;
        if (fresh || !fast || !W.heapFlagClean) HIPCHECK(hipMemsetAsync(W.heapFlag.p, 0, 16 * sizeof(int), st));
        W.heapFlagClean = fast;   // (the frame epilogue clears it again)
    }
    int *overflowFlag = heap ? W.heapFlag.p : reinterpret_cast<int *>(s->counters.p + 2 * C_COUNT) + 12;   // (else: a spare counter word, never set)

    // Enqueue one chunk of `Pc` paths starting at `pathBase`: raygen, R+2 rounds of (intersect, shade), compose.
    // Intersect launch #k traces the closest-hit rays of generation k together with the shadow rays of generation k-1
    // (both exist once k_shade has looked at the hits of generation k-1); k_shade #k then shades generation k-1 with the
    // shadow answers and turns the hits of generation k into shadow rays and the rays of generation k+1.
    bool startEvent = fast;   // the frame's first raygen launch carries its start event (fast frames put nothing but kernels on the stream)
    // (sampleOut: where the chunk's quantised colours go -- the context's sample buffer, or a quadrant level's colour array; epi: the
    // frame epilogue this chunk's compose kernel carries, if any)
    auto enqueue_chunk = [&](const RayGenParams &gp, int *cnt, unsigned *q, int Pc, long long pathBase, uint32_t *sampleOut = nullptr,
                             const FrameEpilogue *epi = nullptr) -> int {
        int *scnt = cnt + (R + 2), *hcnt = cnt + 2 * (R + 2), *acnt = cnt + 3 * (R + 2);   // (acnt[k]: shadow rays of generation k that were really emitted, ShadeArgs::ae)
        const int chunkRow0 = F.stampRows;
        // "long ray first" (kernels.hip): the producer of generation k lists its long rays, launch #k takes them first
        const bool feedback = fast && s->deepMeshes && s->costMap.p != nullptr;
        const bool listLong = s->heavyPath > 0.0f || feedback;
        // (generation 0 and 1 and the shadow rays of generation 0 are the big coherent populations; after two bounces the 64 rays
        // of a packet have little in common and a few packets take three times as long as the rest of their launch: bit 4)
        auto packet_closest = [&](int k) { return (k == 0 ? (pkMask & 1) : (k == 1 ? (pkMask & 4) : ((pkMask & 4) && (pkMask & 16)))) != 0; };
        const bool hinted = fast && nParts == 1 && s->genKey == firstPaths * 64 + nL && !s->noGridHints;
        auto hint = [&](const long long *v, int k) -> long long { return (hinted && k < 68 && v[k] >= 0) ? 4 * v[k] + 4096 : -1; };
        auto packet_shadow = [&](int k) { return (pkMask & 2) != 0 && (k <= 1 || (pkMask & 16) != 0); };   // shadow rays of generation k-1
        auto heavy_for = [&](int k) {
            HeavyArgs H;
            if (listLong && (k == 0 || !heap) && !packet_closest(k)) {   // (packets are not scheduled ray by ray)
                H.list = W.heavyList.p; H.count = hcnt + k; H.path = s->heavyPath;
                if (feedback) { H.costMap = s->costMap.p + (size_t)k * (size_t)framePaths + (size_t)partStart; H.epoch = s->epoch & 0xffffu; H.costThreshold = s->costT[k]; }
            }
            return H;
        };
        // cnt[0] counts the primary rays that reach the scene's root box; index0 lists them
        { Range r("xrt raygen"); launch_raygen(gp, S, rays[0], W.lvlB.p, W.index0.p, cnt, Pc, pathBase, heavy_for(0), st, startEvent ? e0 : nullptr, (int)rayCap); startEvent = false; }
        // (one buffer serves every generation: the closest-hit answers of launch #k are read by part A of k_shade #k alone -- part B works from the
        // slot records -- and launch #k+1 starts after it on the frame's stream)
        xrt_hit *hitsOf[2] = {W.hits.p, W.hits.p};
        int *flagsOf[2] = {W.hitFlags0.p, W.hitFlags0.p};
        SlotRec *slotOf[2] = {W.slot0.p, W.slot1.p};
        int *slotNodeOf[2] = {W.slotNode0.p, W.slotNode1.p};
        for (int k = 0; k <= R + 1; k++) {
            const int cur = k & 1, prv = cur ^ 1;
            const bool hasClosest = k <= R, hasShadow = k >= 1 && nL > 0;
            // a reflection chain keeps the ray of generation k at its parent's slot: their number is scnt[k-1]
            const int *nClosest = (k == 0 || heap || ae) ? cnt + k : scnt + (k - 1);   // (ae: the reflections that were emitted are a compact list, counted by part A)
            IntersectArgs C, B;   // closest-hit segment, shadow segment
            C.rays = rays[cur]; C.hits = hitsOf[cur]; C.index = nullptr; C.nDev = nClosest; C.nMul = 1; C.n = Pc;   // (generation 0: cnt[0] live rays, compact)
            C.nCap = (int)rayCap;
            int *const shadowFlagsOf[2] = {W.shadowFlags.p, ae ? W.shadowFlags1.p : W.shadowFlags.p};   // (the words of generation g: [g & 1])
            C.flags = flagsOf[cur]; B.flags = shadowFlagsOf[(k + 1) & 1];   // launch #k traces the shadow rays of generation k - 1
            C.missRecords = (feedback && !packet_closest(k)) ? 1 : 0;   // k_shade #k reads the cost word of every ray of the generation
            { const HeavyArgs H = heavy_for(k); C.heavyIdx = H.list; C.nHeavy = H.count; }
            B.rays = W.shadowRays.p; B.hits = W.shadowHits.p; B.index = nullptr; B.nDev = hasShadow ? scnt + (k - 1) : nullptr; B.nMul = nL; B.n = 0;
            if (ae && hasShadow) { B.nDev = acnt + (k - 1); B.nMul = 1; B.scatter = W.shadowOut.p; }   // the emitted shadow rays, compact; answers go back to (slot, light)
            B.nCap = (int)((long long)shadowCap * nL);
            for (IntersectArgs *a : {&C, &B}) {
                a->queue = q + QW * k; a->mode = s->sceneMode; a->meshId = 0;
                a->refillMin = s->tune[0]; a->nodeBurst = s->tune[1]; a->leafBurst = s->tune[2]; a->coopMax = s->tune[3]; a->batchMax = s->batchMax; a->heavyShift = s->heavyShift; a->batchMin = s->batchMin; a->spreadMin = s->spreadMin; a->firstBatch = s->firstBatch;
            }
            // segments of coherent rays go to the wave-packet kernel, the others (together, one launch) to the per-lane kernel
            const bool pkC = hasClosest && packet_closest(k), pkB = hasShadow && packet_shadow(k);
            // (I2: a second ray population for the same launch -- the shadow rays beside the closest-hit rays -- or null)
            auto launch_pk = [&](const IntersectArgs &I, int word, long long nHost, const IntersectArgs *I2 = nullptr) -> int {
                PacketArgs PA;
                PA.rays = I.rays; PA.hits = I.hits; PA.flags = I.flags; PA.index = I.index; PA.nDev = I.nDev; PA.nMul = I.nMul; PA.n = I.n; PA.nCap = I.nCap;
                PA.scatter = I.scatter;
                if (I2) { PA.rays2 = I2->rays; PA.hits2 = I2->hits; PA.flags2 = I2->flags; PA.nDev2 = I2->nDev; PA.nMul2 = I2->nMul; PA.nCap2 = I2->nCap; PA.scatter2 = I2->scatter; }
                if (tileCostDev) {   // which tile pays for a packet: the path of its first ray (closest-hit rays: the path list; shadow rays: their hit's slot record)
                    PA.tileCost = tileCostDev; PA.tileBase = (int)pathBase; PA.tileShift = 9 + (gp.samples == 16 ? 4 : (gp.samples == 4 ? 2 : 0));
                    if (&I == &B) { PA.slotOf1 = slotOf[prv]; PA.nL1 = nL; }
                    else PA.pathOf1 = k == 0 ? W.index0.p : paths[cur];
                    if (I2) { PA.slotOf2 = slotOf[prv]; PA.nL2 = nL; }
                }
                PA.queue = q + QW * k + 1 + PACKET_QUEUE_WORDS * word; PA.mode = s->sceneMode; PA.meshId = 0; PA.unmark = 0; PA.staticDiv = s->packetStaticDiv; PA.grabMax = s->packetGrabMax; PA.cullMin = s->packetCullMin; PA.bundle = s->packetBundle ? 1 : 0;
                if (s->packetSplit && s->sceneMode != MODE_SCENE) {
                    if ((rc = split_arena(s, W.splitItems, W.splitRecs, PA, st))) return rc;
                    if (fast && !adaptive && !heap && s->packetLongUs > 0) {   // plain frames: the packets of launch #k are the same from frame to frame while the camera stands still, and nearly so while it moves
                        const size_t stride = (rayCap + 63) / 64 + ((size_t)shadowCap * (size_t)(nL > 0 ? nL : 1) + 63) / 64 + 2;
                        if (W.splitCostStride != stride || !W.splitCost.p) {
                            if ((rc = W.splitCost.ensure(stride * 2 * (size_t)(R + 2)))) return rc;
                            HIPCHECK(hipMemsetAsync(W.splitCost.p, 0, stride * 2 * (size_t)(R + 2) * sizeof(unsigned), st));
                            W.splitCostStride = stride;
                        }
                        PA.splitCost = W.splitCost.p + stride * (size_t)(2 * k + word);
                        PA.splitLong = s->packetLongUs; PA.splitBudgetLong = std::max(1, s->packetBudgetLongUs);
                    }
                }
                hipEvent_t a0 = get_event(F.events, ev), a1 = get_event(F.events, ev + 1);
                if (!a0 || !a1) return fail(XRT_E_HIP, "hipEventCreate failed");
                int grid = s->numCUs * s->blocksPerCUPacket;
                if (nHost >= 0) { const long long want = (nHost + 255) / 256; if (want < grid) grid = (int)(want < 1 ? 1 : want); }
                // a small launch -- a tile shard of a frame, a late generation -- walks parts of the octree no other wave keeps warm: it prefetches (packet.hip pk_prefetch);
                // a launch of many packets per wave has its neighbours for that and would only pay for the extra loads (C5's primary launch: +4 %)
                PA.prefetch = s->packetPrefetch >= 0 ? s->packetPrefetch : ((nHost >= 0 && nHost / 64 < (long long)s->packetPrefetchBelow * grid * 4) ? 1 : 0);
                if (useStamps && grid * 4 <= STAMP_SLOTS && F.stampRows < s->maxStampRows) { PA.stamps = W.stamps.p + (size_t)F.stampRows++ * STAMP_STRIDE; a0 = a1 = nullptr; }
                else if (s->noLaunchTiming) a0 = a1 = nullptr;
                else { pairs.push_back({ev, ev + 1}); ev += 2; }
                launch_packet(S, PA, grid, st, a0, a1);
                return XRT_OK;
            };
            Range ri("xrt intersect #%d", k);
            // both populations of a step in ONE packet launch where both go to the packet kernel: one launch's tail instead of two
            if (pkC && pkB && s->packetMerge) { if ((rc = launch_pk(C, 0, hint(s->genRays, k), &B))) return rc; }
            else {
                if (pkC && (rc = launch_pk(C, 0, k == 0 ? Pc : hint(s->genRays, k)))) return rc;
                if (pkB && (rc = launch_pk(B, 1, hint(s->genRays, k)))) return rc;
            }
            const bool laneC = hasClosest && !pkC, laneB = hasShadow && !pkB;
            if (laneC || laneB) {
                IntersectArgs A = laneC ? C : B;
                if (laneC && laneB) { A.rays2 = B.rays; A.hits2 = B.hits; A.flags2 = B.flags; A.nDev2 = B.nDev; A.nMul2 = B.nMul; A.nCap2 = B.nCap; A.scatter2 = B.scatter; }
                hipEvent_t a0 = get_event(F.events, ev), a1 = get_event(F.events, ev + 1);
                if (!a0 || !a1) return fail(XRT_E_HIP, "hipEventCreate failed");
                const int grid = k == 0 ? persistent_grid(s, Pc) : persistent_grid(s, hint(s->genRays, k), 16);
                if (useStamps && grid * 4 <= STAMP_SLOTS && F.stampRows < s->maxStampRows) { A.stamps = W.stamps.p + (size_t)F.stampRows++ * STAMP_STRIDE; a0 = a1 = nullptr; }
                else if (s->noLaunchTiming) a0 = a1 = nullptr;
                else { pairs.push_back({ev, ev + 1}); ev += 2; }
                if (s->waveTimes.p && k < 16) A.debugTimes = s->waveTimes.p + (size_t)k * 3 * 8192;
                launch_intersect(S, A, s->stackNeeded, grid, st, a0, a1);
            }
            if ((hasClosest || hasShadow) && opts->collect_stats) {   // generation 0: the live list; culled rays are added in frame_finish
                if (hasClosest) launch_count(S, C, s->counters.p, st);
                if (hasShadow) launch_count(S, B, s->counters.p + C_COUNT, st);
            }
            if (ri.on) { roctx().pop(); ri.on = false; }
            Range rs("xrt shade #%d", k);
            ShadeArgs X;
            std::memset(&X, 0, sizeof(X));
            X.level = k; X.doA = hasClosest ? 1 : 0; X.doB = k >= 1 ? 1 : 0;
            X.maxReflections = R; X.P = (int)lvlStride; X.lvl = gp.lvl; X.heap = heap ? 1 : 0; X.overflow = overflowFlag;
            X.rays = rays[cur]; X.hits = hitsOf[cur]; X.hitFlags = flagsOf[cur]; X.shadowFlags = shadowFlagsOf[(k + 1) & 1]; X.nDev = nClosest; X.nHost = Pc; X.cap = (int)rayCap;
            X.index = nullptr; X.rayPath = k == 0 ? W.index0.p : paths[cur];   // (generation 0: the j-th live ray belongs to path index0[j])
            X.rayNode = (heap && k > 0) ? nodesOf[cur] : nullptr; X.rayRef = (heap && k > 0) ? refOf[cur] : nullptr;
            X.slotOut = slotOf[cur]; X.slotNodeOut = heap ? slotNodeOf[cur] : nullptr; X.scnt = scnt + k; X.shadowCap = (int)shadowCap; X.shadowRays = W.shadowRays.p;
            X.nextRays = rays[prv]; X.nextPath = paths[prv]; X.nextNode = heap ? nodesOf[prv] : nullptr; X.nextRef = heap ? refOf[prv] : nullptr;
            X.nextCnt = cnt + k + 1; X.nextCap = (int)rayCap;
            if (ae) { X.ae = 1; X.shadowCnt = acnt + k; X.shadowOut = W.shadowOut.p; X.shadowFlagsOut = shadowFlagsOf[k & 1]; }
            X.slotPrev = slotOf[prv]; X.slotNodePrev = heap ? slotNodeOf[prv] : nullptr; X.scntPrev = k >= 1 ? scnt + (k - 1) : nullptr; X.shadowHits = W.shadowHits.p;
            X.lvlA = W.lvlA.p; X.lvlB = W.lvlB.p; X.lvlAlpha = heap ? W.lvlAlpha.p : nullptr;
            if (k < R) X.heavy = heavy_for(k + 1);
            if (feedback && hasClosest && !packet_closest(k)) { X.costOut = s->costMap.p + (size_t)k * (size_t)framePaths + (size_t)partStart; X.epoch = s->epoch & 0xffffu; }
            {   // (a small generation: 256-thread blocks, so that its work items land on many CUs instead of on the first few)
                const long long h = hint(s->genShade, k);
                if (h >= 0 && h < 4 * 131072 + 4096) launch_shade(S, V, X, st, (int)((h + 255) / 256), 256);
                else launch_shade(S, V, X, st, h < 0 ? 1024 : (int)((h + 1023) / 1024 < 1024 ? (h + 1023) / 1024 : 1024));
            }
        }
        Range rc_("xrt compose");
        StampFold fold;
        fold.src = W.stamps.p; fold.host = F.stampHostDev; fold.row0 = chunkRow0; fold.row1 = F.stampRows;
        if (heap) {
            FrameEpilogue E;
            if (fast) { E.cntSrc = cnt; E.hostCnt = F.pinnedDev; E.cntWords = cntStride; E.zeroWords = cntStride + qStride; E.flagSrc = overflowFlag; }
            launch_compose_tree(W.lvlA.p, W.lvlB.p, W.lvlAlpha.p, Pc, P, R, W.sampleColor.p, wantF32 ? W.sampleF32.p : nullptr, fold, E, st);
        }
        else {
            ResolveArgs RA;
            RA.fused = fuseResolve ? 1 : 0; RA.g = gp; RA.pixelBase = pathBase; RA.out = d_out; RA.outF32 = d_outF32;
            RA.stamps = fold;
            if (epi) { RA.cntSrc = epi->cntSrc; RA.hostCnt = epi->hostCnt; RA.cntWords = epi->cntWords; RA.zeroWords = epi->zeroWords; RA.zeroFrom = epi->zeroFrom; }
            else if (fast && !adaptive) { RA.cntSrc = cnt; RA.hostCnt = F.pinnedDev; RA.cntWords = cntStride; RA.zeroWords = cntStride + qStride; }
            launch_compose(W.lvlA.p, W.lvlB.p, Pc, (int)lvlStride, R, sampleOut ? sampleOut : W.sampleColor.p, (wantF32 && !fuseResolve) ? W.sampleF32.p : nullptr, RA, st,
                           (fast && fuseResolve) ? e1 : nullptr);
        }
        return XRT_OK;
    };
    auto ensure_pinned = [&](size_t bytes) -> int {
        if (F.pinnedBytes < bytes) {
            if (F.pinned) (void)hipHostFree(F.pinned);
            F.pinned = nullptr; F.pinnedBytes = 0;
            HIPCHECK(hipHostMalloc(&F.pinned, bytes + 4096, hipHostMallocMapped));
            F.pinnedBytes = bytes + 4096;
            void *dv = nullptr;
            HIPCHECK(hipHostGetDevicePointer(&dv, F.pinned, 0));
            F.pinnedDev = (int *)dv;
        }
        return XRT_OK;
    };
    auto tally = [&](const int *hc) {
        for (int k = 0; k <= R; k++) {
            shaded += (unsigned long long)hc[(R + 2) + k];
            if (k > 0) closestDeep += (unsigned long long)(heap ? hc[k] : hc[(R + 2) + k - 1]);   // reflection chain: one ray per parent hit
            else live0 += (unsigned long long)hc[0];
        }
    };

    // One pass = trace `total` paths produced by generator `gp`; after every chunk `post(Pc, pathBase)` consumes sampleColor.
    auto run_pass = [&](const RayGenParams &gp, long long total, auto &&post, float progress0, float progress1, bool finalPass) -> int {
        int rc2;
        const size_t nb2 = 2 * C_COUNT * sizeof(unsigned long long);
        if (heap && !fast) {
            // ray-tree mode, the careful way: one chunk at a time, checked for overflow, retried with fewer paths when a generation did not fit
            const size_t words = (size_t)cntStride + (size_t)qStride;
            if ((rc2 = W.cnts.ensure(words)) || (rc2 = ensure_pinned(words * sizeof(int) + nb2 + 64 + 8 + 64))) return rc2;
            W.cntsClean = false;
            unsigned *q = reinterpret_cast<unsigned *>(W.cnts.p + cntStride);
            long long pathBase = 0, curChunk = chunkPaths;   // (ray-tree frames are never split: partStart == 0)
            while (pathBase < total) {
                const int Pc = (int)((total - pathBase) < curChunk ? (total - pathBase) : curChunk);
                HIPCHECK(hipMemsetAsync(W.cnts.p, 0, words * sizeof(int), st));
                const size_t pairsMark = pairs.size(), evMark = ev;
                const int rowsMark = F.stampRows;
                if ((rc2 = enqueue_chunk(gp, W.cnts.p, q, Pc, pathBase))) return rc2;
                // the chunk's samples are consumed before the host has looked at the overflow flag: a retry overwrites them
                if ((rc2 = post(Pc, pathBase))) return rc2;
                char *pin = (char *)F.pinned;
                const size_t cAt = ((size_t)cntStride * sizeof(int) + 7) & ~(size_t)7;   // counters + spare words (overflow flag) in one copy
                HIPCHECK(hipMemcpyAsync(pin, W.cnts.p, (size_t)cntStride * sizeof(int), hipMemcpyDeviceToHost, st));
                HIPCHECK(hipMemcpyAsync(pin + cAt, s->counters.p, nb2, hipMemcpyDeviceToHost, st));
                HIPCHECK(hipMemcpyAsync(pin + cAt + nb2 + 64, overflowFlag, sizeof(int), hipMemcpyDeviceToHost, st));
                HIPCHECK(hipStreamSynchronize(st));
                const int over = *(const int *)(pin + cAt + nb2 + 64);
                if (over) {
                    if (curChunk <= 64) return fail(XRT_E_UNSUPPORTED, "the ray tree of 64 paths does not fit the ray buffers (MaxReflections too high for this scene)");
                    curChunk = (curChunk / 4) & ~63LL;
                    if (curChunk < 64) curChunk = 64;
                    HIPCHECK(hipMemsetAsync(overflowFlag, 0, sizeof(int), st));
                    HIPCHECK(hipMemcpyAsync(s->counters.p, hcntHost, nb2, hipMemcpyHostToDevice, st));   // undo this attempt's counting
                    HIPCHECK(hipStreamSynchronize(st));
                    pairs.resize(pairsMark); ev = evMark; F.stampRows = rowsMark;   // its launches are not part of the frame's timing
                    continue;
                }
                tally((const int *)pin);
                std::memcpy(hcntHost, pin + cAt, nb2);
                pathBase += Pc;
                s->progress.store(progress0 + (progress1 - progress0) * (float)pathBase / (float)total);
            }
            if (finalPass) HIPCHECK(hipEventRecord(e1, st));
            return XRT_OK;
        }
        const int nChunks = (int)((total + chunkPaths - 1) / chunkPaths);
        // ray counts and queue heads of all chunks live in one allocation: one memset per pass
        const size_t cntWords = (size_t)nChunks * cntStride, qWords = (size_t)nChunks * qStride;
        const int *cntsBefore = W.cnts.p;
        if ((rc2 = W.cnts.ensure(cntWords + qWords))) return rc2;
        unsigned *queuesBase = reinterpret_cast<unsigned *>(W.cnts.p + cntWords);
        if (fast && (rc2 = ensure_pinned(cntWords * sizeof(int) + 64))) return rc2;   // (+ the overflow word of a ray-tree frame)
        if (!fast || !W.cntsClean || W.cnts.p != cntsBefore) HIPCHECK(hipMemsetAsync(W.cnts.p, 0, W.cnts.cap * sizeof(int), st));
        W.cntsClean = false;
        for (int c = 0; c < nChunks; c++) {
            const long long localBase = (long long)c * chunkPaths, pathBase = partStart + localBase;
            const int Pc = (int)((total - localBase) < chunkPaths ? (total - localBase) : chunkPaths);
            if ((rc2 = enqueue_chunk(gp, W.cnts.p + (size_t)c * cntStride, queuesBase + (size_t)c * qStride, Pc, pathBase))) return rc2;
            if (!fuseResolve && (rc2 = post(Pc, pathBase))) return rc2;
            if (nChunks > 1) {   // frames of more than MAX_CHUNK_PATHS rays: xrt_progress follows the chunks
                HIPCHECK(hipStreamSynchronize(st));
                s->progress.store(progress0 + (progress1 - progress0) * (float)(c + 1) / (float)nChunks);
            }
        }
        if (fast) {   // counters arrive with k_compose; frame_finish tallies them
            if (!fuseResolve) HIPCHECK(hipEventRecord(e1, st));   // fixed 16 sub-rays: k_resolve is the last kernel
            W.cntsClean = true;
            F.tallyChunks = 1;
            return XRT_OK;
        }
        if (finalPass) HIPCHECK(hipEventRecord(e1, st));
        if (stats) {   // per-pass ray accounting (the counter block is reused by the next pass): one pinned read-back
            const size_t nb = (size_t)nChunks * cntStride * sizeof(int);
            if ((rc2 = ensure_pinned(nb + nb2))) return rc2;
            HIPCHECK(hipMemcpyAsync(F.pinned, W.cnts.p, nb, hipMemcpyDeviceToHost, st));
            HIPCHECK(hipMemcpyAsync((char *)F.pinned + nb, s->counters.p, nb2, hipMemcpyDeviceToHost, st));
            if (finalPass) { F.tallyChunks = nChunks; return XRT_OK; }   // frame_finish tallies after the frame's done event
            HIPCHECK(hipStreamSynchronize(st));
            for (int c = 0; c < nChunks; c++) tally((const int *)F.pinned + (size_t)c * cntStride);
            std::memcpy(hcntHost, (char *)F.pinned + nb, nb2);
        }
        return XRT_OK;
    };

    // valid pixels of this shard
    unsigned long long &validPixels = F.validPixels;
    validPixels = 0;
    const long long slot0 = partStart / (512LL * g.samples), slot1 = (partStart + firstPaths + 512LL * g.samples - 1) / (512LL * g.samples);   // tile slots of this part
    for (long long sl = slot0; sl < slot1; sl++) {
        const long long t = tile_of_slot(sl);
        if (t >= totalTiles) break;
        if (t < 0) continue;   // (an unused slot of a tile table)
        int tx = (int)(t % g.tilesX), ty = (int)(t / g.tilesX);
        int w = g.width - tx * XRT_TILE_W; if (w > XRT_TILE_W) w = XRT_TILE_W;
        int h = g.height - ty * XRT_TILE_H; if (h > XRT_TILE_H) h = XRT_TILE_H;
        validPixels += (unsigned long long)w * h;
    }

    if (!adaptive) {
        livePaths = validPixels * (unsigned long long)g.samples;
        rc = run_pass(g, firstPaths, [&](int Pc, long long pathBase) -> int {
            launch_resolve(g, W.sampleColor.p, wantF32 ? W.sampleF32.p : nullptr, Pc / g.samples, pathBase / g.samples, d_out, d_outF32, st);
            return XRT_OK;
        }, 0.0f, 1.0f, true);
        if (rc != XRT_OK) return rc;
    } else if (F.adaptiveFast) {
        // RenderFirstPass / GetColorForQuadrant (RT:170-311) without a host round trip: every level's pass is enqueued now.  Level 0 has
        // one quadrant per pixel; how many quadrants a deeper level has only the device knows (k_ms_decide counts them into lvlCnt[l],
        // the level's ray generation, compose and fold kernels read that word; the traversal and shading kernels follow the ray
        // counts as always).  Deeper levels get room for one quadrant per pixel (XRT_ADAPTIVE_CAP); a level that needs more sets the
        // overflow word and frame_finish renders the frame again the careful way below.
        const int nPass = quality + 1;
        constexpr int LW = 16;   // words in front of the per-pass counters: lvlCnt[0 .. quality], overflow word at LW - 1
        const long long capDeep = s->adaptiveCap > 0 ? std::min(s->adaptiveCap, totalPixels) : totalPixels;
        auto cap_of = [&](int l) { return l == 0 ? totalPixels : capDeep; };
        const size_t words = (size_t)LW + (size_t)nPass * cntStride + (size_t)nPass * qStride;
        const int *cntsBefore = W.cnts.p;
        if ((rc = W.cnts.ensure(words)) || (rc = ensure_pinned(((size_t)LW + (size_t)nPass * cntStride) * sizeof(int) + 64))) return rc;
        if (!W.cntsClean || W.cnts.p != cntsBefore) HIPCHECK(hipMemsetAsync(W.cnts.p, 0, W.cnts.cap * sizeof(int), st));
        W.cntsClean = false;
        for (int l = 0; l <= quality; l++) {
            auto &Lv = W.levels[l];
            const size_t c = (size_t)cap_of(l);
            if ((rc = Lv.color.ensure(c * 4))) return rc;
            if (l < quality && ((rc = Lv.childBase.ensure(c)) || (rc = Lv.childMask.ensure(c)))) return rc;
            if (l > 0 && ((rc = Lv.cx.ensure(c)) || (rc = Lv.cy.ensure(c)))) return rc;
        }
        int *const lvlCnt = W.cnts.p, *const ovf = W.cnts.p + LW - 1;
        int *const cnt0 = W.cnts.p + LW;
        unsigned *const q0 = reinterpret_cast<unsigned *>(cnt0 + (size_t)nPass * cntStride);
        float size = 1.0f;
        for (int l = 0; l <= quality; l++) {
            auto &Lv = W.levels[l];
            RayGenParams gl = g;
            gl.quadLevel = l; gl.quadSize = size; gl.quadCx = Lv.cx.p; gl.quadCy = Lv.cy.p;
            if (l > 0) { gl.pathsDev = lvlCnt + l; gl.pathsMul = 4; gl.pathsCap = (int)cap_of(l); }
            FrameEpilogue E;
            const bool last = l == quality;
            if (last) { E.cntSrc = W.cnts.p; E.hostCnt = F.pinnedDev; E.cntWords = LW + nPass * cntStride; E.zeroWords = (int)words; E.zeroFrom = LW; }
            if ((rc = enqueue_chunk(gl, cnt0 + (size_t)l * cntStride, q0 + (size_t)l * qStride, (int)(cap_of(l) * 4), 0, Lv.color.p, last ? &E : nullptr))) return rc;
            if (l < quality) {   // RT:279-306
                auto &Nx = W.levels[l + 1];
                launch_ms_decide(gl, Lv.color.p, l > 0 ? lvlCnt + l : nullptr, (int)cap_of(l), 0, Lv.childBase.p, Lv.childMask.p, Nx.cx.p, Nx.cy.p, lvlCnt + l + 1, st,
                                 (int)cap_of(l + 1), ovf);
                size = size / 2.0f;   // RT:290
            }
        }
        for (int l = quality - 1; l >= 0; l--)
            launch_ms_fold(W.levels[l].color.p, W.levels[l + 1].color.p, W.levels[l].childBase.p, W.levels[l].childMask.p, (int)cap_of(l), st, l > 0 ? lvlCnt + l : nullptr);
        RayGenParams g0 = g;
        g0.quadLevel = 0; g0.quadSize = 1.0f;
        launch_resolve(g0, W.levels[0].color.p, nullptr, (int)totalPixels, 0, d_out, d_outF32, st, lvlCnt, LW);   // ... and clears the level words for the next frame
        HIPCHECK(hipEventRecord(e1, st));
        W.cntsClean = true;
        F.tallyChunks = nPass; F.cntBase = LW; F.levelCap = (int)capDeep; F.quality = quality;
    } else {
        // RenderFirstPass / GetColorForQuadrant (RT:170-311), the careful way (a host read-back of every level's size): level 0
        // quadrants are the pixels (size 1); a level's
        // quadrants each cast four rays; corners that deviate are subdivided into the next level, down to
        // MultisampleQuality; results fold back up.
        struct Level { DevBuf<uint32_t> color; DevBuf<int> childBase, childMask; DevBuf<float> cx, cy; long long n = 0; };
        std::vector<Level> lv((size_t)quality + 1);
        int *levelCount = reinterpret_cast<int *>(s->counters.p + 2 * C_COUNT);   // [quality+1] ints in the spare counter words
        auto free_levels = [&]() { for (auto &l : lv) { l.color.release(); l.childBase.release(); l.childMask.release(); l.cx.release(); l.cy.release(); } };
        lv[0].n = totalPixels;
        float size = 1.0f;
        for (int l = 0; l <= quality && rc == XRT_OK; l++) {
            Level &L = lv[l];
            if (L.n == 0) break;
            if ((rc = L.color.ensure((size_t)L.n * 4))) break;
            RayGenParams gl = g;
            gl.quadLevel = l; gl.quadSize = size; gl.quadCx = L.cx.p; gl.quadCy = L.cy.p;
            livePaths += (l == 0 ? validPixels : (unsigned long long)L.n) * 4ull;
            rc = run_pass(gl, L.n * 4, [&](int Pc, long long pathBase) -> int {
                HIPCHECK(hipMemcpyAsync(L.color.p + pathBase, W.sampleColor.p, (size_t)Pc * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
                return XRT_OK;
            }, (float)l / (float)(quality + 1), (float)(l + 1) / (float)(quality + 1), false);
            if (rc != XRT_OK) break;
            if (l < quality) {   // RT:279-306
                Level &N = lv[l + 1];
                if ((rc = L.childBase.ensure((size_t)L.n)) || (rc = L.childMask.ensure((size_t)L.n)) || (rc = N.cx.ensure((size_t)L.n * 4)) ||
                    (rc = N.cy.ensure((size_t)L.n * 4)))
                    break;
                launch_ms_decide(gl, L.color.p, nullptr, (int)L.n, 0, L.childBase.p, L.childMask.p, N.cx.p, N.cy.p, levelCount + l + 1, st);
                int hn = 0;
                hipError_t e = hipMemcpyAsync(&hn, levelCount + l + 1, sizeof(int), hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipStreamSynchronize(st);
                if (e != hipSuccess) { rc = fail(XRT_E_HIP, "adaptive level readback: %s", hipGetErrorString(e)); break; }
                N.n = hn;
                size = size / 2.0f;   // RT:290
            }
        }
        if (rc == XRT_OK) {
            for (int l = quality - 1; l >= 0; l--)
                if (lv[l + 1].n > 0) launch_ms_fold(lv[l].color.p, lv[l + 1].color.p, lv[l].childBase.p, lv[l].childMask.p, (int)lv[l].n, st);
            RayGenParams g0 = g;
            g0.quadLevel = 0; g0.quadSize = 1.0f;
            launch_resolve(g0, lv[0].color.p, nullptr, (int)totalPixels, 0, d_out, d_outF32, st);
            hipError_t e = hipEventRecord(e1, st);
            if (e == hipSuccess && stats) e = hipMemcpyAsync(hcntHost, s->counters.p, sizeof(F.hcnt), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) rc = fail(XRT_E_HIP, "adaptive resolve: %s", hipGetErrorString(e));
        }
        free_levels();
        if (rc != XRT_OK) return rc;
    }
    if (!fast) HIPCHECK(hipEventRecord(F.done, st));
    HIPCHECK(hipGetLastError());
    F.pending = true;
    return XRT_OK;
}

// Waits for a frame enqueued by frame_begin and turns its counters / events into xrt_stats.
int frame_finish(xrt_scene *s, xrt_scene::FrameCtx &F, xrt_stats *stats) {
    if (!F.pending) return fail(XRT_E_INVALID_ARG, "no frame in flight for this ticket");
    F.pending = false;
    HIPCHECK(hipEventSynchronize(F.fast ? F.events[1] : F.done));
    s->progress.store(1.0f);
    if (s->waveTimes.p) {   // development aid: launch k of the frame occupies rows [k*8192, (k+1)*8192) x 3 clocks
        std::vector<unsigned long long> h((size_t)16 * 3 * 8192);
        HIPCHECK(hipMemcpy(h.data(), s->waveTimes.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (FILE *f = fopen(s->waveTimesPath.c_str(), "wb")) { fwrite(h.data(), sizeof(unsigned long long), h.size(), f); fclose(f); }
    }
    if (!s->stampDumpPath.empty() && F.stampRows > 0) {   // development aid (tools/stamp_lives.py)
        std::vector<unsigned long long> h((size_t)F.stampRows * STAMP_STRIDE);
        HIPCHECK(hipMemcpy(h.data(), F.w.stamps.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (FILE *f = fopen(s->stampDumpPath.c_str(), "wb")) { fwrite(h.data(), sizeof(unsigned long long), h.size(), f); fclose(f); }
    }
    const int R = F.R;
    if (F.fast && F.adaptiveFast && ((const int *)F.pinned)[F.cntBase - 1] != 0) {
        // A quadrant level of the adaptive frame did not fit its optimistically sized buffers: the frame is rendered again the careful
        // way (a host read-back per level, exact sizes), and so are this scene's later adaptive frames from the start.
        s->adaptiveFastOk = false;
        F.tallyChunks = 0;
        F.adaptiveFast = false;
        const std::vector<xrt_light> lights = F.redoLights;
        const xrt_camera cam = F.redoCam;
        const xrt_render_opts opts = F.redoOpts;
        int rc = frame_begin(s, F, &cam, lights.data(), (int)lights.size(), &opts, F.redoOut, F.redoOutF32, F.redoSt);
        if (rc != XRT_OK) return rc;
        rc = frame_finish(s, F, stats);
        F.redone = true;
        return rc;
    }
    if (F.fast && F.adaptiveFast) {   // rays of the frame: four per level-0 pixel and four per quadrant of every deeper level
        const int *lw = (const int *)F.pinned;
        F.livePaths = F.validPixels * 4ull;
        for (int l = 1; l <= F.quality; l++) F.livePaths += 4ull * (unsigned long long)std::min(lw[l], F.levelCap);
    }
    if (F.fast && F.heap && F.tallyChunks == 1 && ((const int *)F.pinned)[F.cntStride] != 0) {
        // A generation of the ray tree did not fit the optimistically sized buffers: the frame is rendered again the careful way
        // (chunks, overflow checks, retries with fewer paths), and so are this scene's later ray-tree frames from the start.
        s->heapFastOk = false;
        F.tallyChunks = 0;
        const std::vector<xrt_light> lights = F.redoLights;
        const xrt_camera cam = F.redoCam;
        const xrt_render_opts opts = F.redoOpts;
        int rc = frame_begin(s, F, &cam, lights.data(), (int)lights.size(), &opts, F.redoOut, F.redoOutF32, F.redoSt);
        if (rc != XRT_OK) return rc;
        rc = frame_finish(s, F, stats);
        F.redone = true;
        return rc;
    }
    if (F.tallyChunks > 0) {   // single-pass frame: the read-back was left in flight
        const size_t nb = (size_t)F.tallyChunks * F.cntStride * sizeof(int);
        for (int c = 0; c < F.tallyChunks; c++) {
            const int *hc = (const int *)F.pinned + F.cntBase + (size_t)c * F.cntStride;
            if (hc[0] < 0 || (size_t)hc[0] > F.liveCap)
                return fail(XRT_E_INTERNAL, "%d primary rays reach the scene but the frame's ray arrays were sized for %zu (screen rectangle of the root box)", hc[0], F.liveCap);
            for (int k = 0; k <= R; k++) {
                F.shaded += (unsigned long long)hc[(R + 2) + k];
                if (k > 0) F.closestDeep += (unsigned long long)(F.heap ? hc[k] : hc[(R + 2) + k - 1]);   // reflection chain: one ray per parent hit
                else F.live0 += (unsigned long long)hc[0];
                if (F.ae) {   // queries part A answered itself: the hits' shadow rays that were not emitted, and (not in the last generation) their reflections
                    F.answered += (unsigned long long)hc[(R + 2) + k] * (unsigned long long)F.nL - (unsigned long long)hc[3 * (R + 2) + k];
                    if (k < R) F.answered += (unsigned long long)(hc[(R + 2) + k] - hc[k + 1]);
                }
            }
        }
        if (!F.fast) std::memcpy(F.hcnt, (char *)F.pinned + nb, sizeof(F.hcnt));
        // (adaptive frames in flight put the level-count words in front of the per-pass counters and have no framePaths key: no hints from them)
        if (F.fast && F.tallyChunks == 1 && !F.adaptiveFast) {   // sizes of this frame's generations: grid hints for the next one (sizing only)
            const int *hc = (const int *)F.pinned + F.cntBase;
            for (int k = 0; k <= R + 1 && k < 68; k++) {
                const long long closest = (k == 0 || ((F.heap || F.ae) && k <= R)) ? hc[k] : (k <= R ? hc[(R + 2) + k - 1] : 0), shaded = k >= 1 ? hc[(R + 2) + k - 1] : 0;
                // (a hint shrinks by an eighth per frame at most: a camera that looks away for a frame, or alternates between two views,
                // must not leave the next full view with a grid of sixteen blocks)
                const bool same = s->genKey == F.framePaths * 64 + F.nL;
                const long long rays = closest + shaded * F.nL, work = closest > shaded ? closest : shaded;
                const long long keepR = same && s->genRays[k] > 0 ? s->genRays[k] - s->genRays[k] / 8 : 0, keepS = same && s->genShade[k] > 0 ? s->genShade[k] - s->genShade[k] / 8 : 0;
                s->genRays[k] = rays > keepR ? rays : keepR;
                s->genShade[k] = work > keepS ? work : keepS;
            }
            for (int k = R + 2; k < 68; k++) s->genRays[k] = s->genShade[k] = -1;
            s->genKey = F.framePaths * 64 + F.nL;
        }
        if (F.fast && s->costMap.p && !F.adaptiveFast) {   // steer the "long ray" thresholds towards 2-6 % of each generation's rays
            const int *hc = (const int *)F.pinned + F.cntBase;
            for (int k = 0; k <= R && k < 66; k++) {
                const long long rays = k == 0 ? hc[0] : hc[(R + 2) + k - 1], listed = hc[2 * (R + 2) + k];
                if (rays < 4096) continue;
                if (listed * 100 > rays * s->longFracHi) s->costT[k] = s->costT[k] + s->costT[k] / 4 + 1;
                else if (listed * 100 < rays * s->longFracLo && s->costT[k] > 1) s->costT[k] = s->costT[k] - s->costT[k] / 5 - (s->costT[k] < 5 ? 1 : 0);
                if (s->costT[k] < 1) s->costT[k] = 1;
                if (s->costT[k] > 60000) s->costT[k] = 60000;
            }
        }
        F.tallyChunks = 0;
    }
    {
        float frameMs = 0;
        if (hipEventElapsedTime(&frameMs, F.events[0], F.events[1]) == hipSuccess) s->lastFrameMs = frameMs;
        else (void)hipGetLastError();
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        unsigned long long hcnt[2 * C_COUNT];
        std::memcpy(hcnt, F.hcnt, sizeof(hcnt));
        if (F.collect) {
            // primary rays answered by k_raygen (they miss the scene root box): one query and one OSM:460 test each
            const unsigned long long culled = F.livePaths - F.live0;
            hcnt[C_RAYS] += culled;
            hcnt[C_SCENE_NODES] += culled;
        } else {
            std::memset(hcnt, 0, sizeof(hcnt));
            hcnt[C_RAYS] = F.livePaths + F.closestDeep;
            hcnt[C_HITS] = F.shaded;
            hcnt[C_COUNT + C_RAYS] = F.shaded * (unsigned long long)F.nL;
        }
        fill_stats(stats, hcnt, F.shaded, F.validPixels);
        stats->rays_traversed = stats->rays_closest + stats->rays_shadow - (F.livePaths - F.live0) - F.answered;   // all but the primary rays k_raygen answered and the rays k_shade answered at emission
        if (!F.collect) stats->algorithmic_bytes = 0;   // needs the counting pass
        float ms = 0;
        HIPCHECK(hipEventElapsedTime(&ms, F.events[0], F.events[1]));
        stats->ms_total = ms;
        double mi = 0, longest = 0;
        for (auto &pr : F.pairs) {
            float t = 0;
            HIPCHECK(hipEventElapsedTime(&t, F.events[pr.first], F.events[pr.second]));
            mi += t;
            if (t > longest) longest = t;
        }
        for (int j = 0; j < F.stampRows; j++) {
            const unsigned long long *sp = F.stampHost;
            if (sp[2 * j + 1] > sp[2 * j]) {
                const double t = (double)(sp[2 * j + 1] - sp[2 * j]) / (double)s->wallClockKHz;
                mi += t;
                if (t > longest) longest = t;
            }
        }
        stats->ms_intersect = mi;
        stats->ms_intersect_longest = longest;
        stats->intersect_launches = (uint32_t)F.pairs.size() + (uint32_t)F.stampRows;
        stats->pieces = 1;
    }
    return XRT_OK;
}

// ---- in-library multi-GPU (xrt_render_opts.n_gpus) -----------------------------------------------------------------
constexpr int XRT_MAX_GPUS = 64;

int scene_upload(xrt_scene *scene);

// Copies of the scene on devices device+1 .. device+n-1 (the scene's own device in the single-GPU test mode).
int ensure_replicas(xrt_scene *s, int n) {
    while ((int)s->replicas.size() < n - 1) {
        const int i = (int)s->replicas.size() + 1;
        std::unique_ptr<xrt_scene> r(new xrt_scene());
        r->device = s->fakeGpus ? s->device : s->device + i;
        r->host = s->host;
        r->noRectCull = s->noRectCull; r->oneStream = s->oneStream; r->noFeedback = s->noFeedback; r->overlapMinMs = s->overlapMinMs;
        r->heapRayCap = s->heapRayCap; r->maxChunkPaths = s->maxChunkPaths; r->shadowBytes = s->shadowBytes; r->packetMask = s->packetMask; r->packetMaskHeap = s->packetMaskHeap; r->packetCullMin = s->packetCullMin; r->packetBundle = s->packetBundle; r->packetPrefetch = s->packetPrefetch; r->packetPrefetchBelow = s->packetPrefetchBelow; r->noAnswerAtEmission = s->noAnswerAtEmission; r->packetMerge = s->packetMerge; r->packetSplit = s->packetSplit; r->noLevelMap = s->noLevelMap; r->packetBudgetUs = s->packetBudgetUs; r->packetBudgetItemUs = s->packetBudgetItemUs; r->packetSplitItems = s->packetSplitItems; r->packetLongUs = s->packetLongUs; r->packetBudgetLongUs = s->packetBudgetLongUs; r->batchMax = s->batchMax; r->heavyShift = s->heavyShift; r->heavyShiftGiven = s->heavyShiftGiven; r->batchMin = s->batchMin; r->spreadMin = s->spreadMin; r->tuneGiven = s->tuneGiven;
        for (int k = 0; k < 4; k++) r->tune[k] = s->tune[k];
        HIPCHECK(hipSetDevice(r->device));
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, r->device) == hipSuccess) r->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (hipDeviceGetAttribute(&r->wallClockKHz, hipDeviceAttributeWallClockRate, r->device) != hipSuccess) { r->wallClockKHz = 0; (void)hipGetLastError(); }
        r->launchEvents = s->launchEvents; r->noLaunchTiming = s->noLaunchTiming; r->noGridHints = s->noGridHints; r->heapFastOk = s->heapFastOk; r->maxStampRows = s->maxStampRows; r->adaptiveFastOk = s->adaptiveFastOk; r->adaptiveCap = s->adaptiveCap;
        HIPCHECK(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
        int rc = scene_upload(r.get());
        if (rc != XRT_OK) return rc;
        s->workers.emplace_back(new RankWorker(r->device));
        s->replicas.push_back(r.release());
    }
    return hipSetDevice(s->device) == hipSuccess ? XRT_OK : fail(XRT_E_HIP, "hipSetDevice failed");
}

xrt_scene *rank_scene(xrt_scene *s, int i) { return i == 0 ? s : s->replicas[(size_t)i - 1]; }

// Install / remove a tile table on one scene object (its device must be current).  Caller has checked the table.
int install_tile_table(xrt_scene *r, int w, int h, int count, int tpr, const int *table) {
    if (!table) { r->tileTable.clear(); r->tableW = r->tableH = r->tableCount = r->tableTpr = 0; return XRT_OK; }
    r->tileTable.assign(table, table + (size_t)count * tpr);
    int rc = upload(r->tileTableDev, r->tileTable);
    if (rc != XRT_OK) { r->tileTable.clear(); return rc; }
    r->tableW = w; r->tableH = h; r->tableCount = count; r->tableTpr = tpr;
    return XRT_OK;
}
// The costs of this scene object's tiles, added to cost[tile] (tiles entries); optionally cleared.  Its device must be current, no frame in flight.
int read_tile_costs(xrt_scene *r, int w, int h, float *cost, bool reset) {
    if (r->costW != w || r->costH != h || r->costTiles.empty() || !r->tileCost.p) return XRT_OK;
    std::vector<unsigned> ticks(r->costTiles.size());
    HIPCHECK(hipMemcpy(ticks.data(), r->tileCost.p, ticks.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    const int tiles = ((w + XRT_TILE_W - 1) / XRT_TILE_W) * ((h + XRT_TILE_H - 1) / XRT_TILE_H);
    for (size_t sl = 0; sl < ticks.size(); sl++) { const int t = r->costTiles[sl]; if (t >= 0 && t < tiles) cost[t] += (float)ticks[sl]; }
    if (reset) HIPCHECK(hipMemset(r->tileCost.p, 0, ticks.size() * sizeof(unsigned)));
    return XRT_OK;
}

// Frame `slot` on n devices: rank i renders the tiles t with t % n == i (its own frame context `slot`, its own stream, one
// host thread per device so that frames which need host decisions between passes still run side by side), then ONE
// grouped RCCL exchange moves the tile buffers to rank 0 -- the path's only exchange step -- and k_detile writes the
// W*H frame into d_out on the scene's device.  Everything after the enqueue is stream-ordered on rank 0's stream.
int multi_begin(xrt_scene *s, int slot, const xrt_camera *cam, const xrt_light *lights, int nLights, const xrt_render_opts *opts, uint32_t *d_out,
                hipStream_t *stream0_out) {
    const int n = opts->n_gpus;
    if (opts->shard_count > 1) return fail(XRT_E_INVALID_ARG, "n_gpus > 1 shards the frame inside the library: shard_count must be 0 or 1");
    if (n > XRT_MAX_GPUS) return fail(XRT_E_INVALID_ARG, "n_gpus %d exceeds %d", n, XRT_MAX_GPUS);
    if (!cam || cam->vp_width <= 0 || cam->vp_height <= 0) return fail(XRT_E_INVALID_ARG, "viewport must be positive");
    if (!s->fakeGpus && s->device + n > s->visibleDevices)
        return fail(XRT_E_NO_DEVICE, "n_gpus %d from device %d needs %d visible devices, %d present", n, s->device, s->device + n, s->visibleDevices);
    int rc;
    if ((rc = ensure_replicas(s, n))) return rc;
    {
        std::vector<int> devs;
        if (s->fakeGpus) devs.push_back(s->device);
        else for (int i = 0; i < n; i++) devs.push_back(s->device + i);
        std::string err;
        if (!s->rccl.init(devs, err)) return fail(XRT_E_RCCL, "%s", err.c_str());
        HIPCHECK(hipSetDevice(s->device));
    }
    int tpr = 0, tilesX = 0, tilesY = 0;
    xrt_shard_layout(cam->vp_width, cam->vp_height, n, &tilesX, &tilesY, &tpr);
    // balance_tiles: the tiles are dealt by the last frame's costs (longest first) instead of round-robin -- the static counterpart of the
    // reference's dynamic row stealing (RT:48-52).  The table is made once per frame here and installed on every rank's scene object.
    const bool balanced = opts->balance_tiles != 0 && s->balanceW == cam->vp_width && s->balanceH == cam->vp_height && s->balanceN == n &&
                          (int)s->balanceCost.size() == tilesX * tilesY && !s->frames[slot ^ 1].pending;
    const bool tableFits = s->tableW == cam->vp_width && s->tableH == cam->vp_height && s->tableCount == n && !s->tileTable.empty();
    if (balanced) {
        const int tprB = tpr + (tpr + 3) / 4;
        std::vector<int> table((size_t)n * tprB);
        if ((rc = balance_tiles_impl(tilesX * tilesY, n, s->balanceCost.data(), tprB, table.data()))) return rc;
        for (int i = 0; i < n; i++) {
            xrt_scene *r = rank_scene(s, i);
            HIPCHECK(hipSetDevice(r->device));
            if ((rc = install_tile_table(r, cam->vp_width, cam->vp_height, n, tprB, table.data()))) { (void)hipSetDevice(s->device); return rc; }
        }
        HIPCHECK(hipSetDevice(s->device));
        tpr = tprB;
    } else if (opts->balance_tiles != 0 && tableFits) tpr = s->tableTpr;   // (the other ticket's frame is in flight under the installed table: keep it)
    else if (tableFits && opts->balance_tiles == 0) {   // back to round-robin
        for (int i = 0; i < n; i++) (void)install_tile_table(rank_scene(s, i), 0, 0, 0, 0, nullptr);
    }
    const bool tabled = s->tableW == cam->vp_width && s->tableH == cam->vp_height && s->tableCount == n && !s->tileTable.empty();
    const size_t count = (size_t)tpr * 512;
    if ((rc = s->gathered[slot].ensure(count * (size_t)n))) return rc;
    for (int i = 1; i < n; i++) {
        xrt_scene *r = rank_scene(s, i);
        HIPCHECK(hipSetDevice(r->device));
        if ((rc = r->tileOut[slot].ensure(count))) { (void)hipSetDevice(s->device); return rc; }
        if (s->fakeGpus && !r->tilesReady[slot]) HIPCHECK(hipEventCreateWithFlags(&r->tilesReady[slot], hipEventDisableTiming));
    }
    HIPCHECK(hipSetDevice(s->device));
    // every device's share is enqueued by its own (persistent) host thread
    std::vector<int> rcs((size_t)n, XRT_OK);
    std::vector<std::string> errs((size_t)n);
    auto work_body = [&](int i) {
        xrt_scene *r = rank_scene(s, i);
        if (hipSetDevice(r->device) != hipSuccess) { rcs[(size_t)i] = XRT_E_HIP; errs[(size_t)i] = "hipSetDevice failed"; return; }
        xrt_render_opts o = *opts;
        o.n_gpus = 0; o.shard_rank = i; o.shard_count = n;
        uint32_t *dst = i == 0 ? s->gathered[slot].p : r->tileOut[slot].p;
        rcs[(size_t)i] = frame_begin(r, r->frames[slot], cam, lights, nLights, &o, dst, nullptr, nullptr, 0, 1, false);   // (the gather is enqueued behind the frame: no redo)
        if (rcs[(size_t)i] != XRT_OK) errs[(size_t)i] = g_err;
    };
    // (a rank's job must not throw -- on a worker thread that is std::terminate, on this one it would unwind past workers that still
    // use the locals above: an exception becomes that rank's error code, and every posted worker is waited for whatever happens)
    auto work = [&](int i) {
        try { work_body(i); }
        catch (const std::bad_alloc &) { rcs[(size_t)i] = XRT_E_OOM; errs[(size_t)i] = "out of host memory"; }
        catch (const std::exception &e) { rcs[(size_t)i] = XRT_E_INTERNAL; errs[(size_t)i] = e.what(); }
        catch (...) { rcs[(size_t)i] = XRT_E_INTERNAL; errs[(size_t)i] = "unknown exception"; }
    };
    {
        int posted = 0;
        struct WaitAll {
            xrt_scene *s; int &posted;
            ~WaitAll() { for (int i = 0; i < posted; i++) s->workers[(size_t)i]->wait(); }
        } waitAll{s, posted};
        for (int i = 1; i < n; i++) { s->workers[(size_t)i - 1]->post([&work, i] { work(i); }); posted = i; }
        work(0);
    }
    // From here on frames are in flight on the ranks: whatever fails, every rank's frame is waited for before the error is
    // reported -- a ticket that was never handed out must not leave a context pending (later renders would be XRT_E_BUSY).
    auto drain = [&]() {
        for (int j = 0; j < n; j++) {
            xrt_scene *r = rank_scene(s, j);
            if (r->frames[slot].pending) { (void)hipSetDevice(r->device); (void)frame_finish(r, r->frames[slot], nullptr); }
        }
        (void)hipSetDevice(s->device);
    };
    auto tail = [&]() -> int {
        HIPCHECK(hipSetDevice(s->device));
        for (int i = 0; i < n; i++)
            if (rcs[(size_t)i] != XRT_OK) return fail(rcs[(size_t)i], "rank %d: %s", i, errs[(size_t)i].c_str());
        hipStream_t st0 = s->frames[slot].w.lastStream;
        std::vector<const void *> src;
        std::vector<int> srcRank;
        std::vector<hipStream_t> srcStream;
        std::vector<void *> dst;
        for (int i = 1; i < n; i++) {
            xrt_scene *r = rank_scene(s, i);
            hipStream_t sti = r->frames[slot].w.lastStream;
            if (s->fakeGpus) {   // same device, one communicator: rank 0's stream waits for the tiles, then sends to itself
                HIPCHECK(hipEventRecord(r->tilesReady[slot], sti));
                HIPCHECK(hipStreamWaitEvent(st0, r->tilesReady[slot], 0));
                sti = st0;
            }
            src.push_back(r->tileOut[slot].p); srcRank.push_back(s->fakeGpus ? 0 : i); srcStream.push_back(sti);
            dst.push_back(s->gathered[slot].p + (size_t)i * count);
        }
        {
            std::string err;
            if (!s->rccl.gather(src, srcRank, srcStream, dst, count, st0, err)) return fail(XRT_E_RCCL, "%s", err.c_str());
        }
        HIPCHECK(hipSetDevice(s->device));
        launch_detile(cam->vp_width, cam->vp_height, n, tpr, s->gathered[slot].p, (long long)count, d_out, st0, tabled ? s->tileTableDev.p : nullptr);
        HIPCHECK(hipGetLastError());
        *stream0_out = st0;
        return XRT_OK;
    };
    rc = tail();
    if (rc != XRT_OK) { const std::string keep = g_err; drain(); g_err = keep; }
    return rc;
}

// Counters add up over the pieces of a frame (ranks, halves); times are the slowest piece's (the frame's critical path).
void add_stats(xrt_stats &acc, const xrt_stats &st, bool first) {
    uint64_t *a = reinterpret_cast<uint64_t *>(&acc);
    const uint64_t *b = reinterpret_cast<const uint64_t *>(&st);
    for (size_t k = 0; k < offsetof(xrt_stats, ms_total) / sizeof(uint64_t); k++) a[k] += b[k];
    if (st.ms_total > acc.ms_total) acc.ms_total = st.ms_total;
    if (st.ms_intersect > acc.ms_intersect) acc.ms_intersect = st.ms_intersect;
    if (st.ms_intersect_longest > acc.ms_intersect_longest) acc.ms_intersect_longest = st.ms_intersect_longest;
    if (first) acc.intersect_launches = st.intersect_launches;
    acc.pieces += 1;
    acc.rays_traversed += st.rays_traversed;
    acc.mesh_queries_facing_away += st.mesh_queries_facing_away;
}

int multi_end(xrt_scene *s, int slot, int n, xrt_stats *stats, bool balance) {
    int rc = XRT_OK;
    xrt_stats acc;
    std::memset(&acc, 0, sizeof(acc));
    for (int i = 0; i < n; i++) {
        xrt_scene *r = rank_scene(s, i);
        if (hipSetDevice(r->device) != hipSuccess) { rc = fail(XRT_E_HIP, "hipSetDevice failed"); continue; }
        xrt_stats st;
        std::memset(&st, 0, sizeof(st));
        const int rci = frame_finish(r, r->frames[slot], &st);
        if (rci != XRT_OK) { rc = rci; continue; }
        add_stats(acc, st, i == 0);
    }
    if (rc == XRT_OK && balance && !s->frames[slot ^ 1].pending) {   // this frame's tile costs, all ranks: what the next frame's table is made from
        const xrt_scene::FrameCtx &F0 = s->frames[slot];
        const int w = F0.frameW, h = F0.frameH;
        const int tiles = ((w + XRT_TILE_W - 1) / XRT_TILE_W) * ((h + XRT_TILE_H - 1) / XRT_TILE_H);
        std::vector<float> cost((size_t)tiles, 0.0f);
        bool ok = w > 0 && h > 0;
        for (int i = 0; ok && i < n; i++) {
            xrt_scene *r = rank_scene(s, i);
            ok = hipSetDevice(r->device) == hipSuccess && read_tile_costs(r, w, h, cost.data(), true) == XRT_OK;
        }
        if (ok) { s->balanceCost.swap(cost); s->balanceW = w; s->balanceH = h; s->balanceN = n; }
    }
    (void)hipSetDevice(s->device);
    if (rc == XRT_OK && stats) *stats = acc;
    return rc;
}

// ---- one ticket: frame (on one or n GPUs) -> optional copy into the host's Color[] (RT:122-123) -------------------------
// d_out: the W*H frame in HBM (or this process's tile shard), or null when the frame is only wanted on the host (the
// library then keeps it in a buffer of the ticket).  host_out: page-locked or pageable host memory, or null.
int open_frame_impl(xrt_scene *s, int slot, const xrt_camera *cam, const xrt_light *lights, int nLights, const xrt_render_opts *opts, uint32_t *d_out,
                    float *d_outF32, uint32_t *host_out, hipStream_t st) {
    if (!cam || !opts) return fail(XRT_E_INVALID_ARG, "xrt_render: null argument");
    Range rf("xrt frame (ticket %d)", slot);
    if (opts->n_gpus < 0) return fail(XRT_E_INVALID_ARG, "n_gpus must not be negative");
    const int n = opts->n_gpus > 1 ? opts->n_gpus : 1;
    if (n > 1 && d_outF32) return fail(XRT_E_UNSUPPORTED, "rgb_f32_out with n_gpus > 1");
    const size_t px = (size_t)(cam->vp_width > 0 ? cam->vp_width : 0) * (size_t)(cam->vp_height > 0 ? cam->vp_height : 0);
    int rc;
    if (!d_out) {
        if (opts->shard_count > 1) return fail(XRT_E_INVALID_ARG, "a host frame is a whole frame; use xrt_render_device for shards");
        if ((rc = s->frameOut[slot].ensure(px ? px : 1))) return rc;
        d_out = s->frameOut[slot].p;
    }
    hipStream_t st0 = nullptr;
    int nParts = 1;
    if (n == 1) {
        // Two halves on two streams?  Only plain single-pass frames that run long enough for the drain of their launches to
        // matter, on streams of the library's choosing; by default only when no other frame is in flight to fill the gaps.
        const bool plain = opts->use_multisampling != XRT_MS_ADAPTIVE && !(s->host->arrays.anyTransparent && opts->max_reflections > 0) && !opts->collect_stats;
        const long long px64 = (long long)px * (opts->use_multisampling == XRT_MS_FIXED16 ? 16 : 1) / (opts->shard_count > 1 ? opts->shard_count : 1);
        const bool alone = !s->frames[slot ^ 1].pending;
        if (!st && plain && !s->oneStream && s->splitMode > 0 && (s->splitGiven || s->sceneMode == MODE_SCENE) && (s->splitMode == 2 || alone) && s->lastFrameMs >= s->splitMinMs &&
            s->lastFrameMs >= s->overlapMinMs && px64 >= 8 * 8192 && px64 <= (long long)s->maxChunkPaths)
            nParts = s->splitParts;
        for (int j = 0; j < nParts; j++)
            if ((rc = frame_begin(s, s->frames[slot + 2 * j], cam, lights, nLights, opts, d_out, d_outF32, st, j, nParts))) {
                for (int i = 0; i < j; i++) (void)frame_finish(s, s->frames[slot + 2 * i], nullptr);
                return rc;
            }
        st0 = s->frames[slot].w.lastStream;
    } else if ((rc = multi_begin(s, slot, cam, lights, nLights, opts, d_out, &st0))) return rc;
    xrt_scene::OpenFrame &O = s->open[slot];
    O.nGpus = n; O.balance = n > 1 && opts->balance_tiles != 0;
    O.nParts = nParts;
    O.tail = n > 1 || host_out != nullptr;
    O.hostOut = host_out; O.devOut = d_out; O.px = px; O.st0 = st0;
    if (O.tail) {   // work enqueued behind the frame's own kernels: its end is an event of its own
        for (int j = 1; j < nParts; j++) {   // (the other half ends with an event on its last kernel)
            xrt_scene::FrameCtx &Fj = s->frames[slot + 2 * j];
            HIPCHECK(hipStreamWaitEvent(st0, Fj.fast ? Fj.events[1] : Fj.done, 0));
        }
        if (host_out && px) HIPCHECK(hipMemcpyAsync(host_out, d_out, px * sizeof(uint32_t), hipMemcpyDeviceToHost, st0));   // CurrentTarget.SetData (RT:123)
        if (!s->tailDone[slot]) HIPCHECK(hipEventCreateWithFlags(&s->tailDone[slot], hipEventDisableTiming));
        HIPCHECK(hipEventRecord(s->tailDone[slot], st0));
    }
    return XRT_OK;
}

int close_frame_impl(xrt_scene *s, int slot, xrt_stats *stats) {
    xrt_scene::OpenFrame &O = s->open[slot];
    int rc = XRT_OK;
    if (O.nGpus > 1) rc = multi_end(s, slot, O.nGpus, stats, O.balance);
    else if (O.nParts <= 1) {
        rc = frame_finish(s, s->frames[slot], stats);
        if (rc == XRT_OK && s->frames[slot].redone && O.tail && O.hostOut && O.px) {   // the copy enqueued behind the first attempt took the wrong pixels
            hipError_t e = hipEventSynchronize(s->tailDone[slot]);
            if (e == hipSuccess) e = hipMemcpyAsync(O.hostOut, O.devOut, O.px * sizeof(uint32_t), hipMemcpyDeviceToHost, s->frames[slot].w.lastStream);
            if (e == hipSuccess) e = hipEventRecord(s->tailDone[slot], s->frames[slot].w.lastStream);
            if (e != hipSuccess) rc = fail(XRT_E_HIP, "redo of a ray-tree frame: %s", hipGetErrorString(e));
        }
    }
    else {
        xrt_stats acc;
        std::memset(&acc, 0, sizeof(acc));
        for (int j = 0; j < O.nParts; j++) {
            xrt_stats st;
            std::memset(&st, 0, sizeof(st));
            const int rcj = frame_finish(s, s->frames[slot + 2 * j], &st);
            if (rcj != XRT_OK) rc = rcj;
            else add_stats(acc, st, j == 0);
        }
        if (rc == XRT_OK && stats) *stats = acc;
    }
    if (O.tail) {
        const hipError_t e = hipEventSynchronize(s->tailDone[slot]);
        if (e != hipSuccess && rc == XRT_OK) rc = fail(XRT_E_HIP, "hipEventSynchronize: %s", hipGetErrorString(e));
    }
    O = xrt_scene::OpenFrame();
    if (rc == XRT_OK) rc = guards_check("end of frame");
    return rc;
}

// Split walks (packet.hip): the arena of a context's packet launches and this launch's control words (they sit behind the queue heads and are cleared with them).
// The arena is cleared once, when it is made: an item counts as written when its first word holds the launch's serial number.
int split_arena(xrt_scene *s, DevBuf<unsigned> &items, DevBuf<unsigned> &recs, PacketArgs &PA, hipStream_t st) {
    int rc;
    const size_t NI = ((size_t)s->packetSplitItems + 7) / 8 * 8, NR = NI / 4 + 1;   // (an eighth of the items per XCD)
    if (!items.p || items.cap < NI * SPLIT_ITEM_WORDS) {
        if ((rc = items.ensure(NI * SPLIT_ITEM_WORDS)) || (rc = recs.ensure(NR * SPLIT_REC_WORDS))) return rc;
        HIPCHECK(hipMemsetAsync(items.p, 0, NI * SPLIT_ITEM_WORDS * sizeof(unsigned), st));
        HIPCHECK(hipMemsetAsync(recs.p, 0, NR * SPLIT_REC_WORDS * sizeof(unsigned), st));
    }
    PA.splitItems = items.p; PA.splitRecs = recs.p; PA.splitNI = (int)NI; PA.splitNR = (int)NR;
    if (++s->splitSerial == 0u) s->splitSerial = 1u;
    PA.splitSerial = s->splitSerial;
    PA.splitBudget = std::max(1, s->packetBudgetUs * 100); PA.splitBudgetItem = std::max(1, s->packetBudgetItemUs * 100);   // ticks of the 100 MHz device clock (0 would mean "off")
    PA.splitCtl = reinterpret_cast<unsigned *>((reinterpret_cast<uintptr_t>(PA.queue + PACKET_QUEUE_HEADS * PACKET_HEAD_STRIDE) + 127) & ~(uintptr_t)127);
    return XRT_OK;
}

// The work-queue word of launches on `st` (launches of one stream are ordered, so they can share a word; launches on
// different streams may overlap and must not).  Caller holds apiMutex.
int queue_word_for(xrt_scene *s, hipStream_t st, unsigned **word) {
    constexpr size_t MAX_STREAMS = 1024;
    int rc;
    if ((rc = s->queues.ensure(MAX_STREAMS * (1 + PACKET_QUEUE_WORDS)))) return rc;
    auto it = s->queueOfStream.find(st);
    if (it == s->queueOfStream.end()) {
        if (s->queueOfStream.size() >= MAX_STREAMS) return fail(XRT_E_UNSUPPORTED, "xrt_scene_intersect_device: more than %zu distinct streams on one scene", MAX_STREAMS);
        it = s->queueOfStream.emplace(st, (int)s->queueOfStream.size()).first;
    }
    *word = s->queues.p + (size_t)it->second * (1 + PACKET_QUEUE_WORDS);   // word 0: k_intersect's queue head, then k_packet's heads
    return XRT_OK;
}

// Caller holds apiMutex.
int run_intersect(xrt_scene *s, const xrt_ray *d_rays, int64_t n, xrt_hit *d_hits, int mode, int meshId, hipStream_t st, xrt_stats *stats,
                  bool sync) {
    if (n > 0x7fffffff / 2) return fail(XRT_E_INVALID_ARG, "too many rays in one call");
    int rc;
    unsigned *queue = nullptr;
    if ((rc = queue_word_for(s, st, &queue)) || (rc = s->counters.ensure(2 * C_COUNT + 8))) return rc;
    // the reference-work counters are shared with the frames' counting pass: exact counts need the scene to itself
    if (stats && in_flight(s)) return fail(XRT_E_BUSY, "xrt_scene_intersect with stats while a render is in flight");
    HIPCHECK(hipMemsetAsync(queue, 0, (1 + PACKET_QUEUE_WORDS) * sizeof(unsigned), st));
    IntersectArgs A;
    A.rays = d_rays; A.hits = d_hits; A.index = nullptr; A.nDev = nullptr; A.nMul = 1; A.n = (int)n; A.nCap = 0; A.queue = queue; A.mode = mode; A.meshId = meshId;
    A.refillMin = s->tune[0]; A.nodeBurst = s->tune[1]; A.leafBurst = s->tune[2]; A.coopMax = s->tune[3]; A.batchMax = s->batchMax; A.heavyShift = s->heavyShift; A.batchMin = s->batchMin; A.spreadMin = s->spreadMin; A.firstBatch = s->firstBatch;
    hipEvent_t a0 = nullptr, a1 = nullptr;
    if (stats) {
        a0 = get_event(s, 0); a1 = get_event(s, 1);
        if (!a0 || !a1) return fail(XRT_E_HIP, "hipEventCreate failed");
        HIPCHECK(hipMemsetAsync(s->counters.p, 0, 2 * C_COUNT * sizeof(unsigned long long), st));
    }
    const bool meshOk = mode != MODE_MESH || (meshId >= 0 && meshId < (int)s->host->meshTrees.size() && !s->host->meshTrees[(size_t)meshId].rootIsLeaf);
    if (n > 0 && s->packetMask >= 0 && (s->packetMask & 8) && packet_supported(mode, s->host->arrays.meshDepth, s->host->arrays.sceneDepth) && meshOk) {   // (testing aid: arbitrary batches through the packet kernel)
        PacketArgs PA;
        PA.rays = d_rays; PA.hits = d_hits; PA.n = (int)n; PA.queue = queue + 1; PA.mode = mode; PA.meshId = meshId; PA.bundle = s->packetBundle ? 1 : 0;
        if (s->packetSplit && mode != MODE_SCENE) {
            auto &ar = s->apiSplit[(int)((queue - s->queues.p) / (1 + PACKET_QUEUE_WORDS))];
            if ((rc = split_arena(s, ar.first, ar.second, PA, st))) return rc;
        }
        int grid = s->numCUs * packet_blocks_per_cu(mode);
        const long long want = (n + 255) / 256;
        if (want < grid) grid = (int)want;
        launch_packet(s->view, PA, grid, st, a0, a1);
    } else if (n > 0) launch_intersect(s->view, A, s->stackNeeded, persistent_grid(s, n), st, a0, a1);
    if (stats) {
        if (n > 0) launch_count(s->view, A, s->counters.p, st);
    }
    HIPCHECK(hipGetLastError());
    if (sync || stats) HIPCHECK(hipStreamSynchronize(st));
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        unsigned long long hcnt[2 * C_COUNT];
        HIPCHECK(hipMemcpy(hcnt, s->counters.p, sizeof(hcnt), hipMemcpyDeviceToHost));
        fill_stats(stats, hcnt, 0, 0);
        stats->rays_traversed = stats->rays_closest + stats->rays_shadow;
        float ms = 0;
        if (n > 0) HIPCHECK(hipEventElapsedTime(&ms, a0, a1));
        stats->ms_total = ms; stats->ms_intersect = ms; stats->ms_intersect_longest = ms; stats->intersect_launches = n > 0 ? 1 : 0;
    }
    return XRT_OK;
}

// Host arrays of scene->host -> HBM of scene->device, launch geometry and scheduling defaults (second half of
// xrt_scene_build; also what puts a replica of the scene on another device).
int scene_upload(xrt_scene *scene) {
    const SceneArrays &A = scene->host->arrays;
    scene->stackNeeded = (A.sceneDepth + 1) + (A.meshDepth + 1);
    if (intersect_stack_capacity(scene->stackNeeded) < 0) return fail(XRT_E_UNSUPPORTED, "octree too deep for the LDS stack (%d levels)", scene->stackNeeded);
    if (scene->device < 0) return XRT_OK;   // host-only scene: trees can be inspected, nothing can be traced
    HIPCHECK(hipSetDevice(scene->device));
    int rc;
    if ((rc = upload(scene->blocks, A.blocks)) || (rc = upload(scene->leafNB, A.leafNB)) || (rc = upload(scene->leafTB, A.leafTB)) || (rc = upload(scene->refT, A.refT)) || (rc = upload(scene->pblocks, A.pblocks)) || (rc = upload(scene->lrec, A.lrec)) || (rc = upload(scene->refN, A.refN)) || (rc = upload(scene->refG, A.refG)) ||
        (rc = upload(scene->snodes, A.snodes)) || (rc = upload(scene->shade, A.shade)) || (rc = upload(scene->childDfs, A.childDfs)) ||
        (rc = upload(scene->srefs, A.srefs)) || (rc = upload(scene->scull, A.scull)) || (rc = upload(scene->runTB, A.runTB)) || (rc = upload(scene->triTB, A.triTB)) || (rc = upload(scene->runBase, A.runBase)) || (rc = upload(scene->objMesh, A.objMesh)) ||
        (rc = upload(scene->meshes, A.meshes)) || (rc = upload(scene->objects, A.objects)) || (rc = upload(scene->materials, A.materials)) ||
        (rc = upload(scene->texels, A.texels)))
        return rc;
    SceneView &S = scene->view;
    S.blocks = scene->blocks.p; S.childDfs = scene->childDfs.p; S.leafNB = scene->leafNB.p; S.leafTB = scene->leafTB.p; S.refT = scene->refT.p; S.pblocks = scene->pblocks.p; S.lrec = scene->lrec.p; S.refN = scene->refN.p; S.refG = scene->refG.p;
    S.meshes = scene->meshes.p; S.snodes = scene->snodes.p; S.srefs = scene->srefs.p; S.scull = scene->scull.p; S.runTB = scene->runTB.p; S.triTB = scene->triTB.p; S.runBase = scene->runBase.p;
    S.objects = scene->objects.p; S.objMesh = scene->objMesh.p;
    S.nMeshes = (int)scene->host->meshes.size(); S.nObjects = (int)scene->host->objects.size();
    S.sceneDepth = A.sceneDepth + 1; S.meshDepth = A.meshDepth + 1;
    S.nodeCull = 1;
    if (const char *e = getenv("XRT_NODE_CULL")) { const int v = atoi(e); if (v >= 0 && v <= 2) S.nodeCull = v; }   // (tools: a scheduling-free switch, results never change)
    scene->sceneMode = (scene->host->objects.size() == 1 && scene->host->objects[0].meshes.size() == 1 && scene->host->meshes.size() == 1 &&
                        scene->host->sceneTree.nodeCount == 1) ? MODE_SINGLE : MODE_SCENE;
    if (getenv("XRT_NO_SINGLE")) scene->sceneMode = MODE_SCENE;
    scene->blocksPerCU = intersect_blocks_per_cu(scene->stackNeeded, scene->sceneMode);
    scene->blocksPerCUMesh = intersect_blocks_per_cu(scene->stackNeeded, MODE_MESH);
    scene->packetOk = packet_supported(scene->sceneMode, A.meshDepth, A.sceneDepth);
    // Refill threshold of k_intersect.  A two-level scene refills a wave only when ALL its lanes are idle: fresh rays start in the
    // scene phase while the others are deep in a mesh, and every partial refill made the wave run that phase for a few lanes
    // (measured with 8x8-pixel waves, whose rays take about equally long: C3 3.1 -> 2.45 ms, C4 8.9 -> 6.2 ms of traversal per frame).
    // One-body scenes have no such phase and keep refilling at 24 idle lanes (64 costs them 6 %).
    if (!scene->tuneGiven) scene->tune[0] = scene->sceneMode == MODE_SCENE ? 64 : 24;
    // Listed long rays: mixed one in eight into the first batches where waves refill lane by lane (a wave full of them takes five
    // times as long as one of them: C5 at one sample per pixel 1.58 -> 1.42 ms); 64 to a wave where waves refill as a whole -- there
    // a mixed wave idles 56 lanes until its long rays are done (C3 2.71 -> 2.97 ms when mixed).
    if (!scene->heavyShiftGiven) scene->heavyShift = scene->sceneMode == MODE_SCENE ? 0 : 3;
    scene->blocksPerCUPacket = packet_blocks_per_cu(scene->sceneMode);
    scene->firstBatch = (A.meshDepth == 0) ? 256 : 64;   // every mesh is a single leaf: rays are cheap, avoid queue traffic
    if (const char *e = getenv("XRT_FIRST_BATCH")) { int v = atoi(e); if (v >= 64 && v <= 4096 && v % 64 == 0) scene->firstBatch = v; }
    {   // "long ray first": worth it only where rays can be long, i.e. where some mesh has a real octree
        float frac = 0.25f;
        if (const char *e = getenv("XRT_HEAVY")) frac = (float)atof(e);
        scene->heavyPath = 0.0f;
        scene->deepMeshes = A.meshDepth > 0;
        for (int &t : scene->costT) t = 24;
        if (const char *e = getenv("XRT_LONG_FRAC")) { int lo = 0, hi = 0; if (sscanf(e, "%d,%d", &lo, &hi) == 2 && lo >= 0 && hi > lo && hi <= 100) { scene->longFracLo = lo; scene->longFracHi = hi; } }
        // (measured: +24 % on the 1M-triangle heightfield, whose stragglers are rays skimming the terrain; nothing on the
        //  instanced grid, whose rays are all about as long as the box -- XRT_HEAVY forces it on for any scene)
        const bool wanted = getenv("XRT_HEAVY") != nullptr || scene->sceneMode == MODE_SINGLE;
        if (wanted && frac > 0.0f && A.meshDepth > 0 && A.snodes.size() >= 2) {
            const f4 lo = A.snodes[0], hi = A.snodes[1];
            const double dx = (double)hi.x - lo.x, dy = (double)hi.y - lo.y, dz = (double)hi.z - lo.z;
            const double diag = std::sqrt(dx * dx + dy * dy + dz * dz);
            if (diag > 0.0 && diag < 1e30) scene->heavyPath = (float)(frac * diag);
        }
    }
    scene->resident = true;
    return XRT_OK;
}

}  // namespace

// Every render entry point funnels through these two: no C++ exception (std::bad_alloc from a host-side vector, ...) crosses the C boundary.
int open_frame(xrt_scene *s, int slot, const xrt_camera *cam, const xrt_light *lights, int nLights, const xrt_render_opts *opts, uint32_t *d_out,
               float *d_outF32, uint32_t *host_out, hipStream_t st) {
    return guarded("xrt_render", [&]() -> int { return open_frame_impl(s, slot, cam, lights, nLights, opts, d_out, d_outF32, host_out, st); });
}
int close_frame(xrt_scene *s, int slot, xrt_stats *stats) {
    return guarded("xrt_render_end", [&]() -> int { return close_frame_impl(s, slot, stats); });
}

extern "C" {

int xrt_version(void) { return XRT_VERSION; }
const char *xrt_last_error(void) { return g_err.c_str(); }

int xrt_device_count(int *count_out) {
    if (!count_out) return fail(XRT_E_INVALID_ARG, "xrt_device_count: null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count_out = 0; (void)hipGetLastError(); return XRT_OK; }
    *count_out = n;
    return XRT_OK;
}

int xrt_scene_create(int device, xrt_scene **scene_out) {
    if (!scene_out) return fail(XRT_E_INVALID_ARG, "xrt_scene_create: null argument");
    *scene_out = nullptr;
    if (const char *e = getenv("XRT_GUARD")) g_guardMode.store(atoi(e) != 0 ? 1 : 0);
    if (device < -1) return fail(XRT_E_INVALID_ARG, "xrt_scene_create: bad device index");
    if (device >= 0) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return fail(XRT_E_NO_DEVICE, "no HIP device visible; libxrt has no CPU execution path"); }
        if (device >= n) return fail(XRT_E_NO_DEVICE, "device %d not present (%d visible)", device, n);
        HIPCHECK(hipSetDevice(device));
    }
    xrt_scene *s = new xrt_scene();
    s->device = device;
    if (device >= 0 && hipGetDeviceCount(&s->visibleDevices) != hipSuccess) { (void)hipGetLastError(); s->visibleDevices = 0; }
    s->waveTimesPath = getenv("XRT_WAVE_TIMES") ? getenv("XRT_WAVE_TIMES") : "";
    s->stampDumpPath = getenv("XRT_STAMP_DUMP") ? getenv("XRT_STAMP_DUMP") : "";
    s->fakeGpus = getenv("XRT_FAKE_GPUS") != nullptr;
    if (const char *e = getenv("XRT_SPREAD_MIN")) { const int v = atoi(e); if (v >= 4 && v <= 64 && v % 4 == 0) s->spreadMin = v; }
    if (const char *e = getenv("XRT_BATCH_MIN")) { const int v = atoi(e); if (v >= 16 && v <= 64 && v % 16 == 0) s->batchMin = v; }
    if (const char *e = getenv("XRT_HEAVY_SHIFT")) { const int v = atoi(e); if (v >= 0 && v <= 6) { s->heavyShift = v; s->heavyShiftGiven = true; } }
    if (const char *e = getenv("XRT_BATCH_MAX")) { const int v = atoi(e); if (v >= 16 && v <= 4096 && v % 16 == 0) s->batchMax = v; }
    if (const char *e = getenv("XRT_PK_SPLIT")) s->packetSplit = atoi(e) != 0;
    if (const char *e = getenv("XRT_LEVEL_MAP")) s->noLevelMap = atoi(e) == 0;
    if (const char *e = getenv("XRT_PK_BUDGET")) { const int v = atoi(e); if (v >= 0 && v <= 1000000) s->packetBudgetUs = v; }   // (0: a walk looks for pending subtrees at every block it enters)
    if (const char *e = getenv("XRT_PK_BUDGET_ITEM")) { const int v = atoi(e); if (v >= 0 && v <= 1000000) s->packetBudgetItemUs = v; }
    if (const char *e = getenv("XRT_PK_LONG")) { const int v = atoi(e); if (v >= 0 && v <= 1000000) s->packetLongUs = v; }
    if (const char *e = getenv("XRT_PK_BUDGET_LONG")) { const int v = atoi(e); if (v >= 0 && v <= 1000000) s->packetBudgetLongUs = v; }
    if (const char *e = getenv("XRT_PK_SPLIT_ITEMS")) { const int v = atoi(e); if (v >= 1 && v <= (1 << 20)) s->packetSplitItems = v; }
    if (const char *e = getenv("XRT_PK_GRAB")) { const int v = atoi(e); if (v >= 1 && v <= 64) s->packetGrabMax = v; }
    if (const char *e = getenv("XRT_PK_STATIC")) { const int v = atoi(e); if (v >= 0 && v <= 64) s->packetStaticDiv = v; }
    if (const char *e = getenv("XRT_PACKET")) { const int v = atoi(e); if (v >= -1 && v <= 31) s->packetMask = v; }
    if (const char *e = getenv("XRT_PACKET_HEAP")) { const int v = atoi(e); if (v >= -1 && v <= 31) s->packetMaskHeap = v; }
    if (const char *e = getenv("XRT_SPLIT")) { const int v = atoi(e); if (v >= 0 && v <= 2) { s->splitMode = v; s->splitGiven = true; } }
    if (const char *e = getenv("XRT_LAUNCH_EVENTS")) s->launchEvents = atoi(e) != 0;
    if (const char *e = getenv("XRT_HEAP_FAST")) s->heapFastOk = atoi(e) != 0;
    if (const char *e = getenv("XRT_ADAPTIVE_FAST")) s->adaptiveFastOk = atoi(e) != 0;
    if (const char *e = getenv("XRT_ADAPTIVE_CAP")) { const long long v = atoll(e); if (v >= 1) s->adaptiveCap = v; }
    if (const char *e = getenv("XRT_GRID_HINTS")) s->noGridHints = atoi(e) == 0;
    if (const char *e = getenv("XRT_STAMP_ROWS")) { const int v = atoi(e); if (v >= 0 && v <= MAX_STAMP_ROWS) s->maxStampRows = v; }
    if (const char *e = getenv("XRT_LAUNCH_TIMING")) s->noLaunchTiming = atoi(e) == 0;
    if (const char *e = getenv("XRT_SPLIT_MS")) s->splitMinMs = (float)atof(e);
    if (const char *e = getenv("XRT_SPLIT_PARTS")) { const int v = atoi(e); if (v >= 2 && v <= 4) s->splitParts = v; }
    if (const char *e = getenv("XRT_PK_BUNDLE")) s->packetBundle = atoi(e) != 0;
    if (const char *e = getenv("XRT_PK_PREFETCH")) { const int v = atoi(e); if (v >= -1 && v <= 1) s->packetPrefetch = v; }
    if (const char *e = getenv("XRT_PK_PREFETCH_BELOW")) { const int v = atoi(e); if (v >= 0 && v <= 100000) s->packetPrefetchBelow = v; }
    if (const char *e = getenv("XRT_PK_CULL_MIN")) s->packetCullMin = atoi(e);
    if (const char *e = getenv("XRT_AE")) s->noAnswerAtEmission = atoi(e) == 0;
    if (const char *e = getenv("XRT_PK_MERGE")) s->packetMerge = atoi(e) != 0;
    if (const char *e = getenv("XRT_LEAF_ORDER")) s->hs.spatialRuns = atoi(e) != 0;   // 0: the references of big leaves in list order (tools: A/B of the storage order, results never change)
#ifdef XRT_DEV   // (make DEV=1) the two margin factors are the only switches that can change a result: below their proven values the skips
                 // are no longer exact.  A shipped library does not read them from the environment of its host process.
    if (const char *e = getenv("XRT_LEAF_CULL")) { const double v = atof(e); if (v >= 0.0 && v <= 1e6) s->hs.leafCullSafety = v; }   // 0 = no tight leaf boxes, 1 = the proven margin
    if (const char *e = getenv("XRT_CULL_SAFETY")) { const double v = atof(e); if (v >= 0.0 && v <= 1e6) s->hs.cullSafety = v; }   // factor S of the object pre-cull margin (below 2 the bound is no longer proven)
#endif
    s->noRectCull = getenv("XRT_NO_RECT_CULL") != nullptr; s->oneStream = getenv("XRT_ONE_STREAM") != nullptr; s->noFeedback = getenv("XRT_NO_FEEDBACK") != nullptr;
    if (const char *e = getenv("XRT_OVERLAP_MS")) s->overlapMinMs = (float)atof(e);   // 0: every single-chunk frame gets its context's stream
    if (const char *e = getenv("XRT_HEAP_RAY_CAP")) { long long v = atoll(e); if (v >= 1024 && v <= HEAP_RAY_CAP) s->heapRayCap = v; }
    if (const char *e = getenv("XRT_SHADOW_BYTES")) { long long v = atoll(e); if (v >= (1LL << 20)) s->shadowBytes = v; }
    if (const char *e = getenv("XRT_CHUNK_PATHS")) { long long v = atoll(e); if (v >= 8192 && v <= MAX_CHUNK_PATHS && v % 8192 == 0) s->maxChunkPaths = v; }
    if (const char *t = getenv("XRT_TUNE")) {   // "refill,nodeBurst,leafBurst[,coopMax]" — scheduling only, never results
        int v[4] = {0, 0, 0, s->tune[3]};
        if (sscanf(t, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) >= 3 && v[0] >= 1 && v[0] <= 64 && v[1] >= 1 && v[2] >= 1 && v[3] >= 0 && v[3] <= 64) {
            for (int i = 0; i < 4; i++) s->tune[i] = v[i];
            s->tuneGiven = true;
        }
    }
    if (device >= 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) s->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (hipDeviceGetAttribute(&s->wallClockKHz, hipDeviceAttributeWallClockRate, device) != hipSuccess) { s->wallClockKHz = 0; (void)hipGetLastError(); }
        hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete s; return fail(XRT_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    }
    *scene_out = s;
    return XRT_OK;
}

int xrt_scene_destroy(xrt_scene *scene) {
    if (!scene) return XRT_OK;
    if (in_flight(scene)) return fail(XRT_E_BUSY, "xrt_scene_destroy: a render is in flight");
    delete scene;
    return XRT_OK;
}

int xrt_scene_add_mesh(xrt_scene *scene, const float *v, const float *n, const float *uv, const float *surf_n, const float *color,
                       int32_t ntri, const xrt_material *material, const float bbox[6], int32_t *mesh_id_out) {
    if (!scene || !mesh_id_out) return fail(XRT_E_INVALID_ARG, "xrt_scene_add_mesh: null argument");
    if (in_flight(scene)) return fail(XRT_E_BUSY, "scene is rendering");
    return guarded("xrt_scene_add_mesh", [&]() -> int {
        std::string err;
        int id = scene->hs.add_mesh(v, n, uv, surf_n, color, ntri, material, bbox, err);
        if (id < 0) return fail(XRT_E_INVALID_ARG, "%s", err.c_str());
        scene->resident = false;
        *mesh_id_out = id;
        return XRT_OK;
    });
}

int xrt_scene_add_object(xrt_scene *scene, const int32_t *mesh_ids, int32_t n_meshes, const float world[16], const float inv_world[16],
                         const float bbox[6], const float world_bbox[6], int32_t *object_id_out) {
    if (!scene || !object_id_out) return fail(XRT_E_INVALID_ARG, "xrt_scene_add_object: null argument");
    if (in_flight(scene)) return fail(XRT_E_BUSY, "scene is rendering");
    return guarded("xrt_scene_add_object", [&]() -> int {
        std::string err;
        int id = scene->hs.add_object(mesh_ids, n_meshes, world, inv_world, bbox, world_bbox, err);
        if (id < 0) return fail(XRT_E_INVALID_ARG, "%s", err.c_str());
        scene->resident = false;
        *object_id_out = id;
        return XRT_OK;
    });
}

int xrt_scene_build(xrt_scene *scene, int32_t mesh_threshold, int32_t scene_threshold) {
    if (!scene) return fail(XRT_E_INVALID_ARG, "xrt_scene_build: null scene");
    if (in_flight(scene)) return fail(XRT_E_BUSY, "scene is rendering");
    return guarded("xrt_scene_build", [&]() -> int {
        std::string err;
        scene->resident = false;
        if (!scene->hs.build(mesh_threshold, scene_threshold, err)) return fail(XRT_E_UNSUPPORTED, "%s", err.c_str());
        scene->workers.clear();
        for (xrt_scene *r : scene->replicas) delete r;   // copies of the previous build on other devices
        scene->replicas.clear();
        return scene_upload(scene);
    });
}

int xrt_scene_save(const xrt_scene *scene, const char *path) {
    if (!scene || !path) return fail(XRT_E_INVALID_ARG, "xrt_scene_save: null argument");
    return guarded("xrt_scene_save", [&]() -> int {
        std::string err;
        if (!scene->hs.save(path, err)) return fail(XRT_E_INVALID_ARG, "%s (%s)", err.c_str(), path);
        return XRT_OK;
    });
}

int xrt_scene_load(int device, const char *path, xrt_scene **scene_out) {
    if (!scene_out || !path) return fail(XRT_E_INVALID_ARG, "xrt_scene_load: null argument");
    int rc = xrt_scene_create(device, scene_out);
    if (rc != XRT_OK) return rc;
    rc = guarded("xrt_scene_load", [&]() -> int {
        std::string err;
        if (!(*scene_out)->hs.load(path, err)) return fail(XRT_E_INVALID_ARG, "%s (%s)", err.c_str(), path);
        return XRT_OK;
    });
    if (rc != XRT_OK) {
        const std::string keep = g_err;
        delete *scene_out;
        *scene_out = nullptr;
        g_err = keep;
    }
    return rc;
}

int xrt_scene_get_tree(const xrt_scene *scene, int32_t mesh_id, xrt_node_info *nodes, int64_t *n_nodes_inout, int32_t *refs,
                       int64_t *n_refs_inout) {
    if (!scene || !n_nodes_inout || !n_refs_inout) return fail(XRT_E_INVALID_ARG, "xrt_scene_get_tree: null argument");
    if (!scene->hs.built) return fail(XRT_E_NOT_BUILT, "xrt_scene_get_tree: call xrt_scene_build first");
    if (mesh_id >= (int)scene->hs.meshTrees.size()) return fail(XRT_E_INVALID_ARG, "xrt_scene_get_tree: unknown mesh id");
    const FlatTree &t = mesh_id < 0 ? scene->hs.sceneTree : scene->hs.meshTrees[mesh_id];
    if (nodes) {
        if (*n_nodes_inout < (int64_t)t.info.size()) return fail(XRT_E_INVALID_ARG, "node array too small");
        std::memcpy(nodes, t.info.data(), t.info.size() * sizeof(xrt_node_info));
    }
    if (refs) {
        if (*n_refs_inout < (int64_t)t.infoRefs.size()) return fail(XRT_E_INVALID_ARG, "ref array too small");
        std::memcpy(refs, t.infoRefs.data(), t.infoRefs.size() * sizeof(int32_t));
    }
    *n_nodes_inout = (int64_t)t.info.size();
    *n_refs_inout = (int64_t)t.infoRefs.size();
    return XRT_OK;
}

int xrt_scene_intersect(xrt_scene *scene, const xrt_ray *rays, const int32_t *ignore_object, int64_t n, xrt_hit *hits_out, xrt_stats *stats_out) {
    (void)ignore_object;   // dead in the reference (OSM:343), SURVEY Q8
    int rc = need_device(scene, "xrt_scene_intersect");
    if (rc != XRT_OK) return rc;
    if (n < 0 || (n > 0 && (!rays || !hits_out))) return fail(XRT_E_INVALID_ARG, "xrt_scene_intersect: null argument");
    std::lock_guard<std::mutex> lock(scene->apiMutex);   // concurrent callers (RT:105-113) take turns on the staging buffers
    if ((rc = scene->apiRays.ensure((size_t)n)) || (rc = scene->apiHits.ensure((size_t)n))) return rc;
    hipStream_t st = scene->stream;
    if (n > 0) HIPCHECK(hipMemcpyAsync(scene->apiRays.p, rays, (size_t)n * sizeof(xrt_ray), hipMemcpyHostToDevice, st));
    if ((rc = run_intersect(scene, scene->apiRays.p, n, scene->apiHits.p, scene->sceneMode, 0, st, stats_out, false))) return rc;
    if (n > 0) HIPCHECK(hipMemcpyAsync(hits_out, scene->apiHits.p, (size_t)n * sizeof(xrt_hit), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    return guards_check("end of a batched query");
}

int xrt_scene_intersect_device(xrt_scene *scene, const void *d_rays, int64_t n, void *d_hits_out, void *stream) {
    int rc = need_device(scene, "xrt_scene_intersect_device");
    if (rc != XRT_OK) return rc;
    if (n < 0 || (n > 0 && (!d_rays || !d_hits_out))) return fail(XRT_E_INVALID_ARG, "xrt_scene_intersect_device: null argument");
    if (((uintptr_t)d_rays & 15) || ((uintptr_t)d_hits_out & 15)) return fail(XRT_E_INVALID_ARG, "device buffers must be 16-byte aligned");
    std::lock_guard<std::mutex> lock(scene->apiMutex);   // (held for the enqueue only: the call is asynchronous)
    return run_intersect(scene, (const xrt_ray *)d_rays, n, (xrt_hit *)d_hits_out, scene->sceneMode, 0, (hipStream_t)stream, nullptr, false);
}

int xrt_mesh_intersect(xrt_scene *scene, int32_t mesh_id, const xrt_ray *rays, int64_t n, xrt_hit *hits_out) {
    int rc = need_device(scene, "xrt_mesh_intersect");
    if (rc != XRT_OK) return rc;
    if (mesh_id < 0 || mesh_id >= (int)scene->hs.meshes.size()) return fail(XRT_E_INVALID_ARG, "xrt_mesh_intersect: unknown mesh id");
    if (n < 0 || (n > 0 && (!rays || !hits_out))) return fail(XRT_E_INVALID_ARG, "xrt_mesh_intersect: null argument");
    std::lock_guard<std::mutex> lock(scene->apiMutex);
    if ((rc = scene->apiRays.ensure((size_t)n)) || (rc = scene->apiHits.ensure((size_t)n))) return rc;
    hipStream_t st = scene->stream;
    if (n > 0) HIPCHECK(hipMemcpyAsync(scene->apiRays.p, rays, (size_t)n * sizeof(xrt_ray), hipMemcpyHostToDevice, st));
    if ((rc = run_intersect(scene, scene->apiRays.p, n, scene->apiHits.p, MODE_MESH, mesh_id, st, nullptr, false))) return rc;
    if (n > 0) HIPCHECK(hipMemcpyAsync(hits_out, scene->apiHits.p, (size_t)n * sizeof(xrt_hit), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    return guards_check("end of a batched query");
}

int xrt_render(xrt_scene *scene, const xrt_camera *camera, const xrt_light *lights, int32_t n_lights, const xrt_render_opts *opts,
               uint32_t *rgba_out, float *rgb_f32_out, xrt_stats *stats_out) {
    int rc = need_device(scene, "xrt_render");
    if (rc != XRT_OK) return rc;
    if (!camera || !opts || !rgba_out) return fail(XRT_E_INVALID_ARG, "xrt_render: null argument");
    if (opts->shard_count > 1) return fail(XRT_E_INVALID_ARG, "xrt_render writes a whole frame; use xrt_render_device for shards");
    BusyGuard guard(scene);
    if (!guard.owned || scene->frames[0].pending || scene->frames[1].pending)
        return fail(XRT_E_BUSY, "Current render operation not finished.");   // RT:62-63
    if (camera->vp_width <= 0 || camera->vp_height <= 0) return fail(XRT_E_INVALID_ARG, "viewport must be positive");
    const size_t px = (size_t)camera->vp_width * (size_t)camera->vp_height;
    if (rgb_f32_out && (rc = scene->outF32.ensure(px * 3))) return rc;
    if ((rc = open_frame(scene, 0, camera, lights, n_lights, opts, nullptr, rgb_f32_out ? scene->outF32.p : nullptr, rgba_out, nullptr))) return rc;
    if ((rc = close_frame(scene, 0, stats_out))) return rc;
    if (rgb_f32_out) HIPCHECK(hipMemcpy(rgb_f32_out, scene->outF32.p, px * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return XRT_OK;
}

int xrt_render_begin(xrt_scene *scene, const xrt_camera *camera, const xrt_light *lights, int32_t n_lights, const xrt_render_opts *opts,
                     uint32_t *rgba_out, int32_t *ticket_out) {
    int rc = need_device(scene, "xrt_render_begin");
    if (rc != XRT_OK) return rc;
    if (!camera || !opts || !rgba_out || !ticket_out) return fail(XRT_E_INVALID_ARG, "xrt_render_begin: null argument");
    BusyGuard guard(scene);
    if (!guard.owned) return fail(XRT_E_BUSY, "Current render operation not finished.");
    const int slot = !scene->frames[0].pending ? 0 : (!scene->frames[1].pending ? 1 : -1);
    if (slot < 0) return fail(XRT_E_BUSY, "two frames are already in flight; call xrt_render_end first");
    if ((rc = open_frame(scene, slot, camera, lights, n_lights, opts, nullptr, nullptr, rgba_out, nullptr))) return rc;
    *ticket_out = slot;
    return XRT_OK;
}

int xrt_render_end(xrt_scene *scene, int32_t ticket, xrt_stats *stats_out) { return xrt_render_device_end(scene, ticket, stats_out); }

int xrt_host_register(void *host_ptr, uint64_t bytes) {
    if (!host_ptr || bytes == 0) return fail(XRT_E_INVALID_ARG, "xrt_host_register: null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return fail(XRT_E_NO_DEVICE, "no HIP device visible"); }
    HIPCHECK(hipHostRegister(host_ptr, (size_t)bytes, hipHostRegisterDefault));
    return XRT_OK;
}

int xrt_host_unregister(void *host_ptr) {
    if (!host_ptr) return fail(XRT_E_INVALID_ARG, "xrt_host_unregister: null argument");
    HIPCHECK(hipHostUnregister(host_ptr));
    return XRT_OK;
}

int xrt_render_device(xrt_scene *scene, const xrt_camera *camera, const xrt_light *lights, int32_t n_lights, const xrt_render_opts *opts,
                      void *d_rgba_out, void *stream, xrt_stats *stats_out) {
    int rc = need_device(scene, "xrt_render_device");
    if (rc != XRT_OK) return rc;
    if (!camera || !opts || !d_rgba_out) return fail(XRT_E_INVALID_ARG, "xrt_render_device: null argument");
    BusyGuard guard(scene);
    if (!guard.owned || scene->frames[0].pending || scene->frames[1].pending) return fail(XRT_E_BUSY, "Current render operation not finished.");
    if ((rc = open_frame(scene, 0, camera, lights, n_lights, opts, (uint32_t *)d_rgba_out, nullptr, nullptr, (hipStream_t)stream))) return rc;
    return close_frame(scene, 0, stats_out);
}

int xrt_render_device_begin(xrt_scene *scene, const xrt_camera *camera, const xrt_light *lights, int32_t n_lights, const xrt_render_opts *opts,
                            void *d_rgba_out, void *stream, int32_t *ticket_out) {
    int rc = need_device(scene, "xrt_render_device_begin");
    if (rc != XRT_OK) return rc;
    if (!camera || !opts || !d_rgba_out || !ticket_out) return fail(XRT_E_INVALID_ARG, "xrt_render_device_begin: null argument");
    BusyGuard guard(scene);
    if (!guard.owned) return fail(XRT_E_BUSY, "Current render operation not finished.");
    const int slot = !scene->frames[0].pending ? 0 : (!scene->frames[1].pending ? 1 : -1);
    if (slot < 0) return fail(XRT_E_BUSY, "two frames are already in flight; call xrt_render_device_end first");
    hipStream_t st = (hipStream_t)stream;
    if ((rc = open_frame(scene, slot, camera, lights, n_lights, opts, (uint32_t *)d_rgba_out, nullptr, nullptr, st))) return rc;
    *ticket_out = slot;
    return XRT_OK;
}

int xrt_render_device_end(xrt_scene *scene, int32_t ticket, xrt_stats *stats_out) {
    int rc = need_device(scene, "xrt_render_device_end");
    if (rc != XRT_OK) return rc;
    if (ticket < 0 || ticket > 1) return fail(XRT_E_INVALID_ARG, "xrt_render_device_end: unknown ticket");
    BusyGuard guard(scene);
    if (!guard.owned) return fail(XRT_E_BUSY, "Current render operation not finished.");
    if (!scene->frames[ticket].pending) return fail(XRT_E_INVALID_ARG, "no frame in flight for this ticket");
    return close_frame(scene, ticket, stats_out);
}

int xrt_shard_layout(int32_t width, int32_t height, int32_t shard_count, int32_t *tiles_x_out, int32_t *tiles_y_out, int32_t *tiles_per_rank_out) {
    if (width <= 0 || height <= 0 || shard_count <= 0) return fail(XRT_E_INVALID_ARG, "xrt_shard_layout: bad argument");
    int tx = (width + XRT_TILE_W - 1) / XRT_TILE_W, ty = (height + XRT_TILE_H - 1) / XRT_TILE_H;
    if (tiles_x_out) *tiles_x_out = tx;
    if (tiles_y_out) *tiles_y_out = ty;
    if (tiles_per_rank_out) *tiles_per_rank_out = (int)shard_tiles_per_rank((long long)tx * ty, shard_count, tx);
    return XRT_OK;
}

int xrt_detile_device(int32_t width, int32_t height, int32_t shard_count, const void *d_gathered, int64_t rank_stride, void *d_rgba_out,
                      void *stream) {
    if (width <= 0 || height <= 0 || shard_count <= 0 || !d_gathered || !d_rgba_out) return fail(XRT_E_INVALID_ARG, "xrt_detile_device: bad argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return fail(XRT_E_NO_DEVICE, "no HIP device visible"); }
    int tpr = 0;
    xrt_shard_layout(width, height, shard_count, nullptr, nullptr, &tpr);
    if (rank_stride < 0 || (rank_stride > 0 && rank_stride < (int64_t)tpr * 512)) return fail(XRT_E_INVALID_ARG, "xrt_detile_device: rank_stride smaller than a rank's tiles");
    launch_detile(width, height, shard_count, tpr, (const uint32_t *)d_gathered, rank_stride > 0 ? rank_stride : (long long)tpr * 512,
                  (uint32_t *)d_rgba_out, (hipStream_t)stream);
    HIPCHECK(hipGetLastError());
    return XRT_OK;
}

int xrt_scene_set_tile_table(xrt_scene *scene, int32_t width, int32_t height, int32_t shard_count, int32_t tiles_per_rank, const int32_t *tile_of_slot) {
    int rc = need_device(scene, "xrt_scene_set_tile_table");
    if (rc != XRT_OK) return rc;
    BusyGuard guard(scene);
    if (!guard.owned || scene->frames[0].pending || scene->frames[1].pending) return fail(XRT_E_BUSY, "Current render operation not finished.");
    return guarded("xrt_scene_set_tile_table", [&]() -> int {
        HIPCHECK(hipSetDevice(scene->device));
        if (!tile_of_slot) return install_tile_table(scene, 0, 0, 0, 0, nullptr);
        if (width <= 0 || height <= 0 || shard_count <= 0 || tiles_per_rank <= 0) return fail(XRT_E_INVALID_ARG, "xrt_scene_set_tile_table: bad argument");
        const int tiles = ((width + XRT_TILE_W - 1) / XRT_TILE_W) * ((height + XRT_TILE_H - 1) / XRT_TILE_H);
        int rc2 = check_tile_table(tiles, shard_count, tiles_per_rank, tile_of_slot);
        if (rc2 != XRT_OK) return rc2;
        return install_tile_table(scene, width, height, shard_count, tiles_per_rank, tile_of_slot);
    });
}

int xrt_scene_tile_costs(xrt_scene *scene, int32_t width, int32_t height, float *cost_out, int32_t reset) {
    int rc = need_device(scene, "xrt_scene_tile_costs");
    if (rc != XRT_OK) return rc;
    if (width <= 0 || height <= 0 || !cost_out) return fail(XRT_E_INVALID_ARG, "xrt_scene_tile_costs: bad argument");
    BusyGuard guard(scene);
    if (!guard.owned || scene->frames[0].pending || scene->frames[1].pending) return fail(XRT_E_BUSY, "Current render operation not finished.");
    const int tiles = ((width + XRT_TILE_W - 1) / XRT_TILE_W) * ((height + XRT_TILE_H - 1) / XRT_TILE_H);
    std::fill(cost_out, cost_out + tiles, 0.0f);
    HIPCHECK(hipSetDevice(scene->device));
    return guarded("xrt_scene_tile_costs", [&]() -> int { return read_tile_costs(scene, width, height, cost_out, reset != 0); });
}

int xrt_balance_tiles(int32_t width, int32_t height, int32_t shard_count, const float *tile_cost, int32_t tiles_per_rank, int32_t *tile_of_slot_out) {
    if (width <= 0 || height <= 0 || shard_count <= 0 || tiles_per_rank <= 0 || !tile_of_slot_out) return fail(XRT_E_INVALID_ARG, "xrt_balance_tiles: bad argument");
    const int tiles = ((width + XRT_TILE_W - 1) / XRT_TILE_W) * ((height + XRT_TILE_H - 1) / XRT_TILE_H);
    return guarded("xrt_balance_tiles", [&]() -> int { return balance_tiles_impl(tiles, shard_count, tile_cost, tiles_per_rank, tile_of_slot_out); });
}

int xrt_detile_table_device(int32_t width, int32_t height, int32_t shard_count, int32_t tiles_per_rank, const void *d_tile_of_slot, const void *d_gathered,
                            int64_t rank_stride, void *d_rgba_out, void *stream) {
    if (width <= 0 || height <= 0 || shard_count <= 0 || tiles_per_rank <= 0 || !d_tile_of_slot || !d_gathered || !d_rgba_out)
        return fail(XRT_E_INVALID_ARG, "xrt_detile_table_device: bad argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return fail(XRT_E_NO_DEVICE, "no HIP device visible"); }
    if (rank_stride < 0 || (rank_stride > 0 && rank_stride < (int64_t)tiles_per_rank * 512)) return fail(XRT_E_INVALID_ARG, "xrt_detile_table_device: rank_stride smaller than a rank's tiles");
    launch_detile(width, height, shard_count, tiles_per_rank, (const uint32_t *)d_gathered, rank_stride > 0 ? rank_stride : (long long)tiles_per_rank * 512,
                  (uint32_t *)d_rgba_out, (hipStream_t)stream, (const int *)d_tile_of_slot);
    HIPCHECK(hipGetLastError());
    return XRT_OK;
}

int xrt_rccl_probe(void) {
    return guarded("xrt_rccl_probe", [&]() -> int {
        RcclGather g;
        std::string err;
        if (!g.probe(err)) return fail(XRT_E_RCCL, "%s", err.c_str());
        return XRT_OK;
    });
}

float xrt_progress(const xrt_scene *scene) { return scene ? scene->progress.load() : 0.0f; }

int xrt_split_stats(xrt_scene *scene, uint64_t out[4], int32_t reset) {
    return guarded("xrt_split_stats", [&]() -> int {
        int rc = need_device(scene, "xrt_split_stats");
        if (rc != XRT_OK) return rc;
        if (!out) return fail(XRT_E_INVALID_ARG, "xrt_split_stats: null argument");
        unsigned long long v[4] = {0, 0, 0, 0};
        HIPCHECK(hipSetDevice(scene->device));
        HIPCHECK(hipDeviceSynchronize());
        if (packet_split_stats(v, reset != 0) != 0) return fail(XRT_E_HIP, "xrt_split_stats: %s", hipGetErrorString(hipGetLastError()));
        for (int i = 0; i < 4; i++) out[i] = v[i];
        return XRT_OK;
    });
}

int xrt_generate_primary_rays(xrt_scene *scene, const xrt_camera *camera, xrt_ray *rays_out) {
    int rc = need_device(scene, "xrt_generate_primary_rays");
    if (rc != XRT_OK) return rc;
    if (!camera || !rays_out) return fail(XRT_E_INVALID_ARG, "xrt_generate_primary_rays: null argument");
    xrt_render_opts o;
    std::memset(&o, 0, sizeof(o));
    RayGenParams g;
    if ((rc = make_raygen(camera, &o, g))) return rc;
    const long long slots = (long long)g.tilesX * g.tilesY * 512;
    if (slots > (1LL << 25)) return fail(XRT_E_INVALID_ARG, "frame too large");
    std::lock_guard<std::mutex> lock(scene->apiMutex);
    if ((rc = scene->apiRays.ensure((size_t)slots))) return rc;
    hipStream_t st = scene->stream;
    launch_raygen(g, scene->view, scene->apiRays.p, nullptr, nullptr, nullptr, (int)slots, 0, HeavyArgs(), st);
    HIPCHECK(hipGetLastError());
    std::vector<xrt_ray> tmp((size_t)slots);
    HIPCHECK(hipMemcpyAsync(tmp.data(), scene->apiRays.p, (size_t)slots * sizeof(xrt_ray), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    for (long long i = 0; i < slots; i++) {   // tile order -> row-major
        long long t = i >> 9;
        int within = (int)(i & 511);
        int wx, wy;
        tile_slot_xy(within, wx, wy);
        int x = (int)(t % g.tilesX) * XRT_TILE_W + wx, y = (int)(t / g.tilesX) * XRT_TILE_H + wy;
        if (x < g.width && y < g.height) rays_out[(size_t)y * g.width + x] = tmp[(size_t)i];
    }
    return XRT_OK;
}

}  // extern "C"

// TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Single-steps the traversal state machine of
// xna-ray-trace_amd/csrc/traverse.h on the CPU, one lane at a time, so the pruned front-to-back walk can
// be checked against the oracle without a GPU (`-m "not gpu"` host-logic tests).  libxrt never links
// this file and has no CPU execution path.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../xna-ray-trace_amd/csrc/scene_host.h"
#include "../../xna-ray-trace_amd/csrc/traverse.h"

using namespace xrt;

struct EmuStack {
    unsigned w[256];
    unsigned get(int i) const { return w[i]; }
    void set(int i, unsigned v) { w[i] = v; }
};

struct emu_scene { HostScene hs; std::string err; };

extern "C" {
emu_scene *emu_create() { return new emu_scene(); }
void emu_destroy(emu_scene *s) { delete s; }
const char *emu_error(emu_scene *s) { return s->err.c_str(); }
int emu_add_mesh(emu_scene *s, const float *v, const float *n, const float *uv, const float *sn, const float *color, int ntri,
                 const xrt_material *m, const float *bbox) { return s->hs.add_mesh(v, n, uv, sn, color, ntri, m, bbox, s->err); }
int emu_add_object(emu_scene *s, const int *ids, int n, const float *w, const float *iw, const float *bb, const float *wbb) {
    return s->hs.add_object(ids, n, w, iw, bb, wbb, s->err);
}
int emu_build(emu_scene *s, int mt, int st) { return s->hs.build(mt, st, s->err) ? 0 : -1; }
// factor S of the object pre-cull margin (scene_host.cpp); set before emu_build
void emu_set_cull_safety(emu_scene *s, double f) { s->hs.cullSafety = f; }
// factor on the tight-leaf-box margin (xrt_core.h leaf_certainly_missed): 0 = off, 1 = the proven bound; set before emu_build
void emu_set_leaf_cull(emu_scene *s, double f) { s->hs.leafCullSafety = f; }
int emu_get_tree(emu_scene *s, int mesh, xrt_node_info *nodes, int64_t *nn, int *refs, int64_t *nr) {
    const FlatTree &t = mesh < 0 ? s->hs.sceneTree : s->hs.meshTrees[mesh];
    if (nodes) std::memcpy(nodes, t.info.data(), t.info.size() * sizeof(xrt_node_info));
    if (refs) std::memcpy(refs, t.infoRefs.data(), t.infoRefs.size() * sizeof(int));
    *nn = (int64_t)t.info.size(); *nr = (int64_t)t.infoRefs.size();
    return 0;
}
void emu_tree_stats(emu_scene *s, int mesh, int *out) {
    const FlatTree &t = s->hs.meshTrees[mesh];
    out[0] = t.nodeCount; out[1] = t.leafCount; out[2] = t.emptyLeaves; out[3] = t.maxDepth; out[4] = t.unsafeNodes; out[5] = t.interiors;
}
// mode 0: scene query, mode 1: mesh query, mode 2: single-object scene prologue.  steps_out (nullable, 3 per ray): scene / node / leaf steps.
static int fast_mode = 1;
void emu_set_fast(int f) { fast_mode = f; }
int emu_intersect(emu_scene *s, int mode, int mesh, const xrt_ray *rays, int64_t n, xrt_hit *hits, int64_t *steps_out) {
    if (!s->hs.built) return -1;
    SceneView S = s->hs.host_view();
    if (S.sceneDepth + S.meshDepth > 256) return -2;
    for (int64_t i = 0; i < n; i++) {
        Lane L;
        std::memset(&L, 0, sizeof(L));
        EmuStack stk;
        lane_begin(L, S, mk(rays[i].o[0], rays[i].o[1], rays[i].o[2]), mk(rays[i].d[0], rays[i].d[1], rays[i].d[2]),
                   rays[i].ignore_mesh, rays[i].ignore_tri, (int)i, mode, mesh);
        int64_t st[3] = {0, 0, 0};
        while (L.state != ST_FINISH) {
            if (L.state == ST_SCENE) { advance_scene(L, S, stk); st[0]++; }
            else if (L.state == ST_NODE) { advance_node(L, S, stk, mode, fast_mode != 0 && L.r.par == 0 && !L.weird); st[1]++; }
            else if (L.state == ST_LEAF) { advance_leaf(L, S); st[2]++; }
        }
        HitOut h = lane_result(L, S, mode);
        xrt_hit &o = hits[i];
        std::memset(&o, 0, sizeof(o));
        o.hit = h.hit; o.object = h.object; o.mesh = h.mesh; o.tri = h.tri; o.leaf = h.leaf;
        o.u = h.u; o.v = h.v; o.d = h.d; o.wx = h.wx; o.wy = h.wy; o.wz = h.wz;
        if (steps_out) { steps_out[3 * i] = st[0]; steps_out[3 * i + 1] = st[1]; steps_out[3 * i + 2] = st[2]; }
    }
    return 0;
}

// Wave-level model of k_intersect's scheduling (development aid): 64 lanes, the same refill rule and phase
// bursts as kernels.hip; reports how many wave-steps each phase takes and how many lanes were active in them.
// out[0..]: refills, refilled lanes, scene steps, scene lanes, node steps, node lanes, leaf steps, leaf lanes,
//           node-pop lanes, node-descend lanes, leaf-geometry lanes, outer iterations
int emu_wave_sim(emu_scene *s, int mode, int mesh, const xrt_ray *rays, int64_t n, int nWaves, int refillMin, int nodeBurst, int leafBurst, int64_t *out) {
    if (!s->hs.built) return -1;
    SceneView S = s->hs.host_view();
    for (int i = 0; i < 16; i++) out[i] = 0;
    const int BATCH = 256;
    std::vector<Lane> L(64);
    std::vector<EmuStack> stk(64);
    int64_t queue = (int64_t)nWaves * BATCH;
    for (int w = 0; w < nWaves; w++) {
        for (auto &l : L) { std::memset(&l, 0, sizeof(Lane)); l.state = ST_IDLE; }
        int64_t batchNext = (int64_t)w * BATCH, batchEnd = std::min<int64_t>(batchNext + BATCH, n);
        bool exhausted = batchNext >= n;
        for (;;) {
            int nIdle = 0;
            for (auto &l : L) nIdle += l.state == ST_IDLE;
            if (nIdle) {
                if (!exhausted && (nIdle >= refillMin || nIdle == 64)) {
                    if (batchNext >= batchEnd) {
                        // dynamic batches: emulate the queue as round-robin over waves
                        batchNext = queue + (int64_t)0; queue += BATCH;
                        batchEnd = std::min<int64_t>(batchNext + BATCH, n);
                        if (batchNext >= n) exhausted = true;
                    }
                    if (!exhausted) {
                        int take = (int)std::min<int64_t>(nIdle, batchEnd - batchNext);
                        int rank = 0;
                        for (auto &l : L) if (l.state == ST_IDLE) {
                            if (rank < take) {
                                const xrt_ray &r = rays[batchNext + rank];
                                lane_begin(l, S, mk(r.o[0], r.o[1], r.o[2]), mk(r.d[0], r.d[1], r.d[2]), r.ignore_mesh, r.ignore_tri, (int)(batchNext + rank), mode, mesh);
                            }
                            rank++;
                        }
                        out[0]++; out[1] += take;
                        batchNext += take;
                    }
                }
                if (exhausted && nIdle == 64) break;
            }
            out[11]++;
            auto any = [&](int st) { for (auto &l : L) if (l.state == st) return true; return false; };
            if (mode == MODE_SCENE) while (any(ST_SCENE)) { out[2]++; for (size_t i = 0; i < 64; i++) if (L[i].state == ST_SCENE) { out[3]++; advance_scene(L[i], S, stk[i]); } }
            for (int it = 0; it < nodeBurst && any(ST_NODE); it++) {
                out[4]++;
                for (size_t i = 0; i < 64; i++) { if (L[i].state == ST_IDLE) out[12]++; if (L[i].state == ST_LEAF) out[13]++; }
                for (size_t i = 0; i < 64; i++) if (L[i].state == ST_NODE) {
                    out[5]++;
                    int sp0 = L[i].sp; bool wasEmpty = L[i].mask == 0;
                    advance_node(L[i], S, stk[i], mode, L[i].r.par == 0 && !L[i].weird);
                    if (wasEmpty) out[8]++; else if (L[i].sp > sp0) out[9]++;
                }
            }
            for (int it = 0; it < leafBurst && any(ST_LEAF); it++) {
                out[6]++;
                for (size_t i = 0; i < 64; i++) { if (L[i].state == ST_IDLE || L[i].state == ST_FINISH) out[14]++; if (L[i].state == ST_NODE) out[15]++; }
                for (size_t i = 0; i < 64; i++) if (L[i].state == ST_LEAF) { out[7]++; int sp = L[i].spec; advance_leaf(L[i], S); if (L[i].spec || sp) out[10]++; }
            }
            for (auto &l : L) if (l.state == ST_FINISH) l.state = ST_IDLE;
        }
    }
    return 0;
}
}

// rccl_gather.h — RCCL (librccl.so, loaded on first use) behind xrt_render_opts.n_gpus: one process drives N GPUs, the
// per-frame exchange step is one grouped send/recv of the image-tile buffers onto the scene's device.
// Only libxrt's in-library multi-GPU path uses this; a single-GPU host never loads RCCL.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

namespace xrt {

class RcclGather {
public:
    ~RcclGather();
    // Communicators for `devices` (distinct HIP device ids, rank i = devices[i]).  Idempotent for the same list.
    bool init(const std::vector<int> &devices, std::string &err);
    int ranks() const { return (int)comms_.size(); }
    // Can librccl be loaded and does it export what the gather needs?  No device is touched (xrt_rccl_probe).
    bool probe(std::string &err) { return load(err); }
    // One grouped exchange: for every i, `count` 32-bit words go from src[i] (on rank srcRank[i], enqueued on
    // srcStream[i]) to dst[i] on rank 0 (enqueued on dstStream).  srcRank[i] == 0 is a send-to-self on rank 0's
    // communicator (RCCL matches the pairs of one group in order) -- the form the single-device test mode uses.
    bool gather(const std::vector<const void *> &src, const std::vector<int> &srcRank, const std::vector<hipStream_t> &srcStream,
                const std::vector<void *> &dst, size_t count, hipStream_t dstStream, std::string &err);

private:
    bool load(std::string &err);
    void destroy();
    void *lib_ = nullptr;
    std::vector<void *> comms_;   // ncclComm_t
    std::vector<int> devices_;
    // the few RCCL entry points used (rccl.h signatures; ncclResult_t / ncclDataType_t are ints, ncclSuccess == 0)
    int (*commInitAll_)(void **, int, const int *) = nullptr;
    int (*commDestroy_)(void *) = nullptr;
    int (*groupStart_)() = nullptr;
    int (*groupEnd_)() = nullptr;
    int (*send_)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*recv_)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*errorString_)(int) = nullptr;
};

}  // namespace xrt

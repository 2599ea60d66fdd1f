// rccl_gather.cpp — see rccl_gather.h.  The signatures follow /opt/rocm/include/rccl/rccl.h (RCCL 2.27): the header is
// included for its types and every entry point is resolved with dlsym, so libxrt.so itself has no load-time dependency
// on librccl (PyTorch-ROCm ships a librccl of the same SONAME; whichever the process loaded first serves both).
#include "rccl_gather.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>

namespace xrt {

static_assert(sizeof(ncclComm_t) == sizeof(void *), "ncclComm_t is an opaque pointer");

bool RcclGather::load(std::string &err) {
    if (lib_) return true;
    // XRT_RCCL_LIB=<path> names the library to load instead (a site with RCCL elsewhere; the CPU test of this failure path)
    const char *given = getenv("XRT_RCCL_LIB");
    const char *defaults[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string why;   // message of the last failed dlopen (dlerror() clears the error it returns: read it once per failure)
    for (const char *n : defaults) {
        if (given) n = given;
        (void)dlerror();
        lib_ = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (lib_) break;
        const char *e = dlerror();
        why = e ? e : "?";
        if (given) break;
    }
    if (!lib_) { err = std::string("cannot load librccl.so: ") + why; return false; }
    auto sym = [&](const char *n) -> void * {
        void *p = dlsym(lib_, n);
        if (!p && err.empty()) err = std::string("librccl.so lacks ") + n;
        return p;
    };
    commInitAll_ = reinterpret_cast<decltype(commInitAll_)>(sym("ncclCommInitAll"));
    commDestroy_ = reinterpret_cast<decltype(commDestroy_)>(sym("ncclCommDestroy"));
    groupStart_ = reinterpret_cast<decltype(groupStart_)>(sym("ncclGroupStart"));
    groupEnd_ = reinterpret_cast<decltype(groupEnd_)>(sym("ncclGroupEnd"));
    send_ = reinterpret_cast<decltype(send_)>(sym("ncclSend"));
    recv_ = reinterpret_cast<decltype(recv_)>(sym("ncclRecv"));
    errorString_ = reinterpret_cast<decltype(errorString_)>(sym("ncclGetErrorString"));
    if (!commInitAll_ || !commDestroy_ || !groupStart_ || !groupEnd_ || !send_ || !recv_ || !errorString_) { lib_ = nullptr; return false; }
    return true;
}

void RcclGather::destroy() {
    for (void *c : comms_) if (c && commDestroy_) (void)commDestroy_(c);
    comms_.clear();
    devices_.clear();
}

RcclGather::~RcclGather() { destroy(); }

bool RcclGather::init(const std::vector<int> &devices, std::string &err) {
    if (!load(err)) return false;
    if (devices == devices_ && !comms_.empty()) return true;
    destroy();
    std::vector<void *> comms(devices.size(), nullptr);
    const int rc = commInitAll_(comms.data(), (int)devices.size(), devices.data());
    if (rc != (int)ncclSuccess) { err = std::string("ncclCommInitAll: ") + errorString_(rc); return false; }
    comms_ = comms;
    devices_ = devices;
    return true;
}

bool RcclGather::gather(const std::vector<const void *> &src, const std::vector<int> &srcRank, const std::vector<hipStream_t> &srcStream,
                        const std::vector<void *> &dst, size_t count, hipStream_t dstStream, std::string &err) {
    if (comms_.empty()) { err = "RCCL communicators not initialised"; return false; }
    int rc = groupStart_();
    for (size_t i = 0; i < src.size() && rc == (int)ncclSuccess; i++) {
        const int r = srcRank[i];
        if (r < 0 || r >= (int)comms_.size()) { (void)groupEnd_(); err = "gather: rank out of range"; return false; }
        rc = send_(src[i], count, (int)ncclUint32, 0, comms_[(size_t)r], srcStream[i]);
        if (rc == (int)ncclSuccess) rc = recv_(dst[i], count, (int)ncclUint32, r, comms_[0], dstStream);
    }
    const int rc2 = groupEnd_();
    if (rc == (int)ncclSuccess) rc = rc2;
    if (rc != (int)ncclSuccess) { err = std::string("RCCL send/recv: ") + errorString_(rc); return false; }
    return true;
}

}  // namespace xrt

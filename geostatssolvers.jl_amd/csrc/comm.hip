// Multi-GPU behind the C-ABI: the preprocess state of a handle (kriging factor, FFTGS spectrum, LUGS factor) reaches
// the other GPUs of the node without any help from the host language beyond carrying a few bytes.
//
// The reference cuts its simulation solvers into `preprocess` (once) and `solvesingle` (mapped over realisations, on
// worker processes if the caller asks: /root/reference/src/simulation/fft.jl:62,145, lu.jl:76,171).  With one process
// per GPU the state computed on rank 0 has to arrive in every peer's HBM; two routes are exported:
//
//   gss_comm_* + gss_state_bcast     RCCL (ncclBroadcast over xGMI) inside the library.  The host carries the 128-byte
//                                    unique id from rank 0 to the peers (Julia: remotecall; Python: any broadcast).
//   gss_state_ipc_export / _import   HIP IPC: the owner exports an 80-byte token, a peer maps the owner's buffer and
//                                    pulls it with one device-to-device copy (its own xGMI link to the owner: seven
//                                    peers pull over seven links at once).  Works between two processes on ONE device
//                                    too, which is how the one-GPU test drives it.
//
// RCCL is loaded at run time (dlopen): a process that never calls gss_comm_init does not pay for it, and a process
// that already holds a copy (torch ships one) shares it.
#include "gss_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

namespace gss {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static ncclComm_t g_comm = nullptr;
static int g_rank = -1, g_nranks = 0;

static int32_t rccl_load() {
  if (g_rccl.lib) return GSS_OK;
  const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
  void* lib = nullptr;
  for (const char* n : names) {
    lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);   // a copy the process already holds
    if (lib) break;
  }
  for (size_t i = 0; !lib && i < sizeof(names) / sizeof(names[0]); ++i) lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!lib) {
    set_error("RCCL is not available: %s", dlerror());
    return GSS_ERR_UNSUPPORTED;
  }
  RcclApi a;
  a.lib = lib;
  a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
  a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
  a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  a.Broadcast = reinterpret_cast<decltype(a.Broadcast)>(dlsym(lib, "ncclBroadcast"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.Broadcast || !a.GetErrorString) {
    set_error("RCCL library lacks an expected symbol");
    return GSS_ERR_UNSUPPORTED;
  }
  g_rccl = a;
  return GSS_OK;
}

#define GSS_NCCL(call)                                                                                  \
  do {                                                                                                  \
    ncclResult_t r__ = (call);                                                                          \
    if (r__ != ncclSuccess) {                                                                           \
      set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(r__));         \
      return GSS_ERR_HIP;                                                                               \
    }                                                                                                   \
  } while (0)

// device address and size of the state of a handle; `adopt` afterwards marks it usable on a receiving handle
static int32_t state_of(int32_t kind, void* handle, void** p, int64_t* bytes) {
  GSS_REQUIRE(handle != nullptr, "NULL handle");
  switch (kind) {
    case GSS_STATE_KRIG: return gss_krig_factor_buffer(static_cast<gss_krig_t*>(handle), p, bytes);
    case GSS_STATE_FFTGS: return gss_fftgs_state_buffer(static_cast<gss_fftgs_t*>(handle), p, bytes);
    case GSS_STATE_LUGS: return gss_lugs_state_buffer(static_cast<gss_lugs_t*>(handle), p, bytes);
    default: set_error("unknown state kind %d", kind); return GSS_ERR_INVALID;
  }
}

static int32_t adopt(int32_t kind, void* handle, void* stream) {
  switch (kind) {
    case GSS_STATE_KRIG: return gss_krig_adopt_factor(static_cast<gss_krig_t*>(handle));
    case GSS_STATE_FFTGS: return gss_fftgs_adopt_state(static_cast<gss_fftgs_t*>(handle), stream);
    default: return gss_lugs_adopt_state(static_cast<gss_lugs_t*>(handle));
  }
}

static int g_last_ipc_route = -1;   // of the last gss_state_ipc_import (gss_stat "ipc_route": tests)
int comm_last_ipc_route() { return g_last_ipc_route; }

struct IpcToken {            // what gss_state_ipc_export writes (GSS_IPC_TOKEN_BYTES)
  hipIpcMemHandle_t mem;     // 64 bytes: the allocation that holds the state
  int64_t offset;            // of the state inside that allocation
  int64_t bytes;
  int32_t pci_domain, pci_bus, pci_device;   // the owner's device: the importer decides between "my own device",
  int32_t reserved;                          // "a peer I can reach" and "a device I cannot reach" before it maps anything
};
static_assert(sizeof(IpcToken) == GSS_IPC_TOKEN_BYTES, "token layout");

static int32_t pci_of(int dev, int32_t* dom, int32_t* bus, int32_t* devid) {
  int a = 0, b = 0, c = 0;
  GSS_HIP(hipDeviceGetAttribute(&a, hipDeviceAttributePciDomainID, dev));
  GSS_HIP(hipDeviceGetAttribute(&b, hipDeviceAttributePciBusId, dev));
  GSS_HIP(hipDeviceGetAttribute(&c, hipDeviceAttributePciDeviceId, dev));
  *dom = a;
  *bus = b;
  *devid = c;
  return GSS_OK;
}

// How the importing process reaches the owner's device.  GSS_IPC_SAME: it is this process's current device (two ranks on
// one GPU, the one-GPU tests); GSS_IPC_PEER: another visible device with peer access -- enabled here, the copy then runs
// over the direct xGMI link; GSS_IPC_HIDDEN: not among the visible devices (one process per GPU behind
// HIP_VISIBLE_DEVICES): the IPC mapping itself is the test, with hipIpcMemLazyEnablePeerAccess; GSS_IPC_NOPEER: visible
// but not reachable -- refused with a message instead of a failing copy.
enum { GSS_IPC_SAME = 0, GSS_IPC_PEER = 1, GSS_IPC_HIDDEN = 2, GSS_IPC_NOPEER = 3 };
static int32_t ipc_route(const IpcToken& t, int* route, int* owner_dev) {
  int cur = 0, ndev = 0;
  GSS_HIP(hipGetDevice(&cur));
  GSS_HIP(hipGetDeviceCount(&ndev));
  *route = GSS_IPC_HIDDEN;
  *owner_dev = -1;
  for (int d = 0; d < ndev; ++d) {
    int32_t dom, bus, dv;
    GSS_TRY(pci_of(d, &dom, &bus, &dv));
    if (dom == t.pci_domain && bus == t.pci_bus && dv == t.pci_device) {
      *owner_dev = d;
      break;
    }
  }
  if (*owner_dev == cur) *route = GSS_IPC_SAME;
  else if (*owner_dev >= 0) {
    int can = 0;
    GSS_HIP(hipDeviceCanAccessPeer(&can, cur, *owner_dev));
    *route = can ? GSS_IPC_PEER : GSS_IPC_NOPEER;
  }
  // test hook (one-GPU box): GSS_IPC_FORCE_ROUTE=peer|hidden|nopeer overrides the decision above
  if (const char* f = std::getenv("GSS_IPC_FORCE_ROUTE")) {
    if (!std::strcmp(f, "peer")) *route = GSS_IPC_PEER;
    else if (!std::strcmp(f, "hidden")) *route = GSS_IPC_HIDDEN;
    else if (!std::strcmp(f, "nopeer")) *route = GSS_IPC_NOPEER;
  }
  return GSS_OK;
}

}  // namespace gss

using namespace gss;

extern "C" {

int32_t gss_comm_unique_id(uint8_t* id) {
  GSS_ENTRY();
  GSS_REQUIRE(id != nullptr, "gss_comm_unique_id: NULL buffer");
  GSS_TRY(rccl_load());
  ncclUniqueId u;
  GSS_NCCL(g_rccl.GetUniqueId(&u));
  static_assert(sizeof(u) == GSS_COMM_ID_BYTES, "unique id size");
  std::memcpy(id, &u, sizeof(u));
  return GSS_OK;
}

int32_t gss_comm_init(const uint8_t* id, int32_t rank, int32_t nranks) {
  GSS_ENTRY();
  GSS_REQUIRE(id != nullptr && nranks >= 1 && rank >= 0 && rank < nranks, "gss_comm_init: bad arguments");
  GSS_REQUIRE(g_comm == nullptr, "gss_comm_init: a communicator exists already (gss_comm_destroy first)");
  GSS_TRY(rccl_load());
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  GSS_NCCL(g_rccl.CommInitRank(&g_comm, nranks, u, rank));   // on the device gss_init bound this process to
  g_rank = rank;
  g_nranks = nranks;
  return GSS_OK;
}

int32_t gss_comm_info(int32_t* rank, int32_t* nranks) {
  GSS_ENTRY();
  if (rank) *rank = g_rank;
  if (nranks) *nranks = g_nranks;
  return GSS_OK;
}

int32_t gss_comm_destroy(void) {
  GSS_ENTRY();
  if (g_comm) {
    GSS_NCCL(g_rccl.CommDestroy(g_comm));
    g_comm = nullptr;
  }
  g_rank = -1;
  g_nranks = 0;
  return GSS_OK;
}

int32_t gss_state_bcast(int32_t kind, void* handle, int32_t root, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(g_comm != nullptr, "gss_state_bcast: no communicator (gss_comm_init)");
  GSS_REQUIRE(root >= 0 && root < g_nranks, "gss_state_bcast: root %d outside 0..%d", root, g_nranks - 1);
  void* p = nullptr;
  int64_t bytes = 0;
  GSS_TRY(state_of(kind, handle, &p, &bytes));   // (on the root this joins an asynchronous fit first)
  hipStream_t s = to_stream(stream);
  GSS_REQUIRE(bytes % 8 == 0, "state size %lld is not a whole number of doubles", (long long)bytes);
  GSS_NCCL(g_rccl.Broadcast(p, p, (size_t)(bytes / 8), ncclFloat64, root, g_comm, s));
  GSS_HIP(hipStreamSynchronize(s));
  if (g_rank != root) GSS_TRY(adopt(kind, handle, stream));
  return GSS_OK;
}

int32_t gss_state_ipc_export(int32_t kind, void* handle, uint8_t* token) {
  GSS_ENTRY();
  GSS_REQUIRE(token != nullptr, "gss_state_ipc_export: NULL token");
  void* p = nullptr;
  int64_t bytes = 0;
  GSS_TRY(state_of(kind, handle, &p, &bytes));
  GSS_HIP(hipDeviceSynchronize());   // whatever still writes the state has finished before a peer may read it
  void* base = nullptr;
  size_t span = 0;
  GSS_HIP(hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&base), &span, p));
  IpcToken t;
  std::memset(&t, 0, sizeof(t));
  GSS_HIP(hipIpcGetMemHandle(&t.mem, base));
  t.offset = (int64_t)(static_cast<char*>(p) - static_cast<char*>(base));
  t.bytes = bytes;
  int cur = 0;
  GSS_HIP(hipGetDevice(&cur));
  GSS_TRY(pci_of(cur, &t.pci_domain, &t.pci_bus, &t.pci_device));
  std::memcpy(token, &t, sizeof(t));
  return GSS_OK;
}

int32_t gss_state_ipc_import(int32_t kind, void* handle, const uint8_t* token, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(token != nullptr, "gss_state_ipc_import: NULL token");
  void* p = nullptr;
  int64_t bytes = 0;
  GSS_TRY(state_of(kind, handle, &p, &bytes));
  IpcToken t;
  std::memcpy(&t, token, sizeof(t));
  GSS_REQUIRE(t.bytes == bytes, "state sizes differ: the owner exports %lld bytes, this handle holds %lld (same grid / "
              "data on both sides?)", (long long)t.bytes, (long long)bytes);
  hipStream_t s = to_stream(stream);
  int route = GSS_IPC_HIDDEN, owner = -1;
  GSS_TRY(ipc_route(t, &route, &owner));
  g_last_ipc_route = route;
  if (route == GSS_IPC_NOPEER) {
    set_error("gss_state_ipc_import: the owner's device (PCI %04x:%02x:%02x) is visible to this process but not peer-accessible "
              "from its current device; use gss_state_bcast (RCCL) or recompute the state on this rank",
              (unsigned)t.pci_domain, (unsigned)t.pci_bus, (unsigned)t.pci_device);
    return GSS_ERR_UNSUPPORTED;
  }
  if (route == GSS_IPC_PEER && owner >= 0) {
    const hipError_t pe = hipDeviceEnablePeerAccess(owner, 0);
    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
      set_error("gss_state_ipc_import: enabling peer access to device %d failed: %s", owner, hipGetErrorString(pe));
      return GSS_ERR_HIP;
    }
    (void)hipGetLastError();   // (hipErrorPeerAccessAlreadyEnabled is sticky otherwise)
  }
  void* src = nullptr;
  {
    const hipError_t oe = hipIpcOpenMemHandle(&src, t.mem, hipIpcMemLazyEnablePeerAccess);
    if (oe != hipSuccess) {
      set_error("gss_state_ipc_import: mapping the owner's buffer (device PCI %04x:%02x:%02x, %s) failed: %s",
                (unsigned)t.pci_domain, (unsigned)t.pci_bus, (unsigned)t.pci_device,
                route == GSS_IPC_HIDDEN ? "not visible to this process" : "visible", hipGetErrorString(oe));
      return GSS_ERR_HIP;
    }
  }
  const hipError_t e = hipMemcpyAsync(p, static_cast<char*>(src) + t.offset, (size_t)bytes, hipMemcpyDeviceToDevice, s);
  const hipError_t e2 = e == hipSuccess ? hipStreamSynchronize(s) : e;
  (void)hipIpcCloseMemHandle(src);
  if (e2 != hipSuccess) {
    set_error("copy from the owner's buffer failed: %s", hipGetErrorString(e2));
    return GSS_ERR_HIP;
  }
  return adopt(kind, handle, stream);
}

}  // extern "C"

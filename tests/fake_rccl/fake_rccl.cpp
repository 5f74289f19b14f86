// fake_rccl.cpp — TEST INFRASTRUCTURE: a stand-in for librccl on a one-GPU box.
//
// libyart_hip.so loads RCCL by name on first use (csrc/multi_device.inc: a table of function pointers); the environment
// variable YART_RCCL_LIB names the library. This stand-in implements exactly the eight entry points that table binds —
// ncclCommInitAll / CommDestroy / CommAbort / GroupStart / GroupEnd / Send / Recv / GetErrorString — with the semantics the
// merge relies on, so that the `distinct == true` branch of the multi-device code (stands for the reference's finishTile
// merge, cpu/tile-renderer.hpp:225-241) runs, and can be made to FAIL, without a second GPU:
//   * "ranks" may share one device (the real library refuses that); a matched send / receive pair becomes a device-to-device
//     hipMemcpyAsync on the receiver's stream, ordered after the sender's stream by an event;
//   * point-to-point operations are only accepted inside a group and are matched at ncclGroupEnd; an operation left
//     unmatched there is reported (the real library would hang: the reason the product must never submit one);
//   * FAKE_RCCL_FAIL_FN = init | send | recv | groupend and FAKE_RCCL_FAIL_CALL = k make the k-th call of that function
//     (counted from 1 over the process) return ncclInternalError;
//   * FAKE_RCCL_LOG = path: one line per call, for the tests to read the ORDER of the product's calls (e.g. that a group is
//     closed on every path, and what happens to communicators aborted while a group is open).
// Nothing in yart_amd/ references this file; it is built by tests (and __graft_entry__.build) into tests/fake_rccl/_build/.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct FakeComm {
  int rank = 0, nranks = 0, device = 0;
  bool aborted = false;
  uint32_t magic = 0xfacec0de;
};

struct Op { bool send; void* buf; size_t bytes; int peer; FakeComm* comm; hipStream_t stream; };

std::mutex g_mu;
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
int g_calls[4] = {0, 0, 0, 0};                 // init, send, recv, groupend

void logLine(const char* fmt, ...) {
  const char* path = std::getenv("FAKE_RCCL_LOG");
  if (!path || !*path) return;
  std::lock_guard<std::mutex> lock(g_mu);
  FILE* f = std::fopen(path, "a");
  if (!f) return;
  va_list ap; va_start(ap, fmt); std::vfprintf(f, fmt, ap); va_end(ap);
  std::fputc('\n', f);
  std::fclose(f);
}

bool injected(int which, const char* name) {
  int n;
  { std::lock_guard<std::mutex> lock(g_mu); n = ++g_calls[which]; }
  const char* fn = std::getenv("FAKE_RCCL_FAIL_FN");
  const char* k = std::getenv("FAKE_RCCL_FAIL_CALL");
  return fn && k && std::strcmp(fn, name) == 0 && std::atoi(k) == n;
}

size_t typeSize(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}

}  // namespace

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
  if (injected(0, "init")) { logLine("ncclCommInitAll n=%d -> INJECTED FAILURE", ndev); return ncclInternalError; }
  if (!comms || ndev <= 0) return ncclInvalidArgument;
  for (int i = 0; i < ndev; i++) {
    FakeComm* c = new FakeComm;
    c->rank = i; c->nranks = ndev; c->device = devlist ? devlist[i] : i;
    comms[i] = reinterpret_cast<ncclComm_t>(c);
  }
  logLine("ncclCommInitAll n=%d", ndev);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  FakeComm* c = reinterpret_cast<FakeComm*>(comm);
  if (!c || c->magic != 0xfacec0de) return ncclInvalidArgument;
  logLine("ncclCommDestroy rank=%d group_depth=%d", c->rank, t_depth);
  c->magic = 0; delete c;
  return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm) {
  FakeComm* c = reinterpret_cast<FakeComm*>(comm);
  if (!c || c->magic != 0xfacec0de) return ncclInvalidArgument;
  size_t queued = 0;
  for (const Op& o : t_ops) if (o.comm == c) queued++;
  logLine("ncclCommAbort rank=%d group_depth=%d queued_ops_on_comm=%zu", c->rank, t_depth, queued);
  // Aborted while a group of this thread still holds operations on it: they are dropped and the communicator stays
  // allocated (marked) until the group closes — the group's list must never point at freed memory.
  if (queued) { c->aborted = true; return ncclSuccess; }
  c->magic = 0; delete c;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() { t_depth++; logLine("ncclGroupStart depth=%d", t_depth); return ncclSuccess; }

static ncclResult_t queue(bool send, void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  FakeComm* c = reinterpret_cast<FakeComm*>(comm);
  if (!c || c->magic != 0xfacec0de) return ncclInvalidArgument;
  if (c->aborted) return ncclInvalidUsage;
  if (t_depth == 0) return ncclInvalidUsage;            // (the product always groups its point-to-point calls)
  if (peer < 0 || peer >= c->nranks || (!buf && count)) return ncclInvalidArgument;
  t_ops.push_back(Op{send, buf, count * typeSize(type), peer, c, stream});
  return ncclSuccess;
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  if (injected(1, "send")) { logLine("ncclSend peer=%d -> INJECTED FAILURE", peer); return ncclInternalError; }
  const ncclResult_t r = queue(true, const_cast<void*>(sendbuff), count, datatype, peer, comm, stream);
  logLine("ncclSend rank=%d peer=%d count=%zu rc=%d", comm ? reinterpret_cast<FakeComm*>(comm)->rank : -1, peer, count, int(r));
  return r;
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  if (injected(2, "recv")) { logLine("ncclRecv peer=%d -> INJECTED FAILURE", peer); return ncclInternalError; }
  const ncclResult_t r = queue(false, recvbuff, count, datatype, peer, comm, stream);
  logLine("ncclRecv rank=%d peer=%d count=%zu rc=%d", comm ? reinterpret_cast<FakeComm*>(comm)->rank : -1, peer, count, int(r));
  return r;
}

ncclResult_t ncclGroupEnd() {
  if (t_depth == 0) return ncclInvalidUsage;
  if (--t_depth > 0) { logLine("ncclGroupEnd depth=%d (inner)", t_depth + 1); return ncclSuccess; }
  std::vector<Op> ops; ops.swap(t_ops);
  const bool fail = injected(3, "groupend");
  ncclResult_t rc = fail ? ncclInternalError : ncclSuccess;
  size_t matched = 0, unmatched = 0, dropped = 0;
  std::vector<bool> used(ops.size(), false);
  std::vector<FakeComm*> reclaim;
  for (size_t i = 0; i < ops.size(); i++) {
    if (ops[i].comm->aborted) { dropped++; used[i] = true; bool seen = false; for (FakeComm* c : reclaim) seen |= c == ops[i].comm; if (!seen) reclaim.push_back(ops[i].comm); }
  }
  for (size_t i = 0; i < ops.size() && !fail; i++) {
    if (used[i] || ops[i].send) continue;
    const Op& r = ops[i];
    size_t j = 0;
    for (; j < ops.size(); j++)
      if (!used[j] && ops[j].send && ops[j].comm->rank == r.peer && ops[j].peer == r.comm->rank && ops[j].comm->nranks == r.comm->nranks) break;
    if (j == ops.size()) continue;
    const Op& s = ops[j];
    used[i] = used[j] = true;
    if (s.bytes != r.bytes) { rc = ncclInvalidArgument; continue; }
    // receiver's stream after the sender's; the copy; sender's stream after the copy (its buffer may be reused then)
    hipEvent_t e0, e1;
    bool ok = hipSetDevice(s.comm->device) == hipSuccess && hipEventCreateWithFlags(&e0, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventRecord(e0, s.stream) == hipSuccess;
    ok = ok && hipSetDevice(r.comm->device) == hipSuccess && hipStreamWaitEvent(r.stream, e0, 0) == hipSuccess;
    ok = ok && hipMemcpyAsync(r.buf, s.buf, r.bytes, hipMemcpyDeviceToDevice, r.stream) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess && hipEventRecord(e1, r.stream) == hipSuccess;
    ok = ok && hipSetDevice(s.comm->device) == hipSuccess && hipStreamWaitEvent(s.stream, e1, 0) == hipSuccess;
    if (ok) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); matched++; }
    else rc = ncclUnhandledCudaError;
  }
  for (size_t i = 0; i < ops.size(); i++) if (!used[i] && !fail) unmatched++;      // (an injected failure submits nothing)
  if (unmatched && rc == ncclSuccess) rc = ncclInvalidUsage;          // (the real library would wait forever for the peer)
  for (FakeComm* c : reclaim) { c->magic = 0; delete c; }
  logLine("ncclGroupEnd ops=%zu matched_pairs=%zu unmatched=%zu dropped_on_aborted_comms=%zu rc=%d%s", ops.size(), matched, unmatched, dropped,
          int(rc), fail ? " (INJECTED FAILURE)" : "");
  return rc;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled hip error (stand-in)";
    case ncclInternalError: return "internal error (stand-in)";
    case ncclInvalidArgument: return "invalid argument (stand-in)";
    case ncclInvalidUsage: return "invalid usage (stand-in)";
    default: return "error (stand-in)";
  }
}

}  // extern "C"

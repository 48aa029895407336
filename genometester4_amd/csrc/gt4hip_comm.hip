/*
 * gt4hip_comm.hip -- the exchange step of the key-range sharded operations: a gatherv
 * of the per-shard record payloads to the writer rank over RCCL (xGMI inside a node).
 *
 * The reference has no counterpart: its multi-list job (scripts/MakeUnion.pl:31-95) exchanges
 * .list files on disk between glistcompare processes.  Here every rank holds one key range of the
 * result in HBM; rank order = key order, so the gathered concatenation is the sorted result.
 * RCCL has no native gatherv (ncclGather takes equal counts, rccl.h:745): it is the grouped
 * point-to-point form -- root: ncclRecv x (G-1) at the offsets the totals give, others: one
 * ncclSend -- on the fully connected node every sender uses its own link to the root.
 *
 * librccl.so is 0.5 GB: it is loaded (dlopen) by the first gt4hip_comm_* call only, so that
 * single-GPU runs of the command-line tool never pay for it.  Host-only code.
 */
#include "gt4hip_host.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

namespace {

struct Rccl {
  void *handle;
  ncclResult_t (*GetUniqueId) (ncclUniqueId *);
  ncclResult_t (*CommInitRank) (ncclComm_t *, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy) (ncclComm_t);
  ncclResult_t (*Send) (const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Recv) (void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*AllGather) (const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  ncclResult_t (*GroupStart) (void);
  ncclResult_t (*GroupEnd) (void);
  const char *(*GetErrorString) (ncclResult_t);
};

Rccl g_rccl;
thread_local char g_comm_err[256] = "";

const Rccl *rccl ()
{
  if (g_rccl.handle) return &g_rccl;
  static const char *const names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
  void *h = NULL;
  /* With NCCL_DEBUG unset RCCL prints a five-line version banner on STDOUT when the first
   * communicator is made; a drop-in tool's stdout must stay the reference's.  NCCL_DEBUG=NONE
   * silences it (measured: tools/rccl_probe.sh); a value the user has set is respected. */
  if (!getenv ("GT4HIP_RCCL_VERBOSE")) setenv ("NCCL_DEBUG", "NONE", 0);
  const char *env = getenv ("GT4HIP_RCCL_LIB");
  if (env) h = dlopen (env, RTLD_NOW | RTLD_GLOBAL);
  for (size_t i = 0; !h && i < sizeof names / sizeof names[0]; i++) h = dlopen (names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    snprintf (g_comm_err, sizeof g_comm_err, "librccl.so could not be loaded: %s", dlerror ());
    return NULL;
  }
  Rccl r;
  r.handle = h;
  r.GetUniqueId = (decltype (r.GetUniqueId)) dlsym (h, "ncclGetUniqueId");
  r.CommInitRank = (decltype (r.CommInitRank)) dlsym (h, "ncclCommInitRank");
  r.CommDestroy = (decltype (r.CommDestroy)) dlsym (h, "ncclCommDestroy");
  r.Send = (decltype (r.Send)) dlsym (h, "ncclSend");
  r.Recv = (decltype (r.Recv)) dlsym (h, "ncclRecv");
  r.AllGather = (decltype (r.AllGather)) dlsym (h, "ncclAllGather");
  r.GroupStart = (decltype (r.GroupStart)) dlsym (h, "ncclGroupStart");
  r.GroupEnd = (decltype (r.GroupEnd)) dlsym (h, "ncclGroupEnd");
  r.GetErrorString = (decltype (r.GetErrorString)) dlsym (h, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Send || !r.Recv || !r.AllGather || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
    snprintf (g_comm_err, sizeof g_comm_err, "librccl.so lacks a required symbol");
    dlclose (h);
    return NULL;
  }
  g_rccl = r;
  return &g_rccl;
}

}  // namespace

struct gt4hip_comm {
  gt4hip_context *ctx;
  ncclComm_t comm;
  int n_ranks, rank;
  unsigned long long *tot_dev;  /* totals exchange: [2] sent + [2 * n_ranks] received, device */
  unsigned long long *tot_host; /* ... the same, pinned */
};

extern "C" const char *gt4hip_comm_last_error (void) { return g_comm_err; }

extern "C" int gt4hip_comm_unique_id (void *id_out)
{
  static_assert (sizeof (ncclUniqueId) == GT4HIP_COMM_ID_BYTES, "id size");
  if (!id_out) return GT4HIP_EINVAL;
  const Rccl *r = rccl ();
  if (!r) return GT4HIP_ECOMM;
  ncclUniqueId id;
  const ncclResult_t e = r->GetUniqueId (&id);
  if (e != ncclSuccess) {
    snprintf (g_comm_err, sizeof g_comm_err, "ncclGetUniqueId: %s", r->GetErrorString (e));
    return GT4HIP_ECOMM;
  }
  memcpy (id_out, &id, sizeof id);
  return GT4HIP_OK;
}

extern "C" int gt4hip_comm_create (gt4hip_context *ctx, const void *id_bytes, int n_ranks, int rank, gt4hip_comm **out)
{
  if (!ctx || !id_bytes || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return GT4HIP_EINVAL;
  const Rccl *r = rccl ();
  if (!r) return gt4hip_fail (ctx, GT4HIP_ECOMM, "%s", g_comm_err);
  HIPCHK (ctx, hipSetDevice (ctx->device));
  gt4hip_comm *c = new (std::nothrow) gt4hip_comm ();
  if (!c) return gt4hip_fail (ctx, GT4HIP_ENOMEM, "host allocation failed");
  c->ctx = ctx;
  c->n_ranks = n_ranks;
  c->rank = rank;
  ncclUniqueId id;
  memcpy (&id, id_bytes, sizeof id);
  const ncclResult_t e = r->CommInitRank (&c->comm, n_ranks, id, rank);
  if (e != ncclSuccess) {
    delete c;
    return gt4hip_fail (ctx, GT4HIP_ECOMM, "ncclCommInitRank (rank %d of %d): %s", rank, n_ranks, r->GetErrorString (e));
  }
  *out = c;
  return GT4HIP_OK;
}

extern "C" void gt4hip_comm_destroy (gt4hip_comm *c)
{
  if (!c) return;
  const Rccl *r = rccl ();
  if (r) {
    hipSetDevice (c->ctx->device);
    hipStreamSynchronize (c->ctx->stream);
    r->CommDestroy (c->comm);
  }
  if (c->tot_dev) hipFree (c->tot_dev);
  if (c->tot_host) hipHostFree (c->tot_host);
  delete c;
}

extern "C" int gt4hip_comm_rank (const gt4hip_comm *c) { return c ? c->rank : -1; }
extern "C" int gt4hip_comm_size (const gt4hip_comm *c) { return c ? c->n_ranks : 0; }

/* The totals exchange of a sharded step (exchange 1): every rank's (n_words, total_count) to every rank --
 * header totals and output offsets -- as ONE ncclAllGather of two 64-bit words per rank on the library's stream, with
 * one stream synchronisation for the whole exchange (the bench's torch.distributed form took a stream synchronisation,
 * a Python all_gather and a device-to-host copy per step: 0.2 - 0.4 ms against a 4 - 5 ms shard merge at 8 GPUs).
 * totals[2 r], totals[2 r + 1] = rank r's pair.  gt4hip_comm_allgather_u64: the same for n <= 8 words per rank (an
 * operation with several outputs exchanges all their totals at once): all[n r + i] = word i of rank r. */
extern "C" int gt4hip_comm_allgather_u64 (gt4hip_comm *c, const uint64_t *mine, uint32_t n, uint64_t *all)
{
  if (!c || !mine || !all || !n || n > 8) return GT4HIP_EINVAL;
  gt4hip_context *ctx = c->ctx;
  const Rccl *r = rccl ();
  if (!r) return gt4hip_fail (ctx, GT4HIP_ECOMM, "%s", g_comm_err);
  HIPCHK (ctx, hipSetDevice (ctx->device));
  const size_t cap = 8 + 8 * (size_t) c->n_ranks; /* room for the widest exchange */
  if (!c->tot_dev) { /* (both or neither: a half-made pair must not survive a failed call) */
    unsigned long long *dev = NULL, *host = NULL;
    HIPCHK (ctx, hipMalloc ((void **) &dev, cap * 8));
    const hipError_t eh = hipHostMalloc ((void **) &host, cap * 8, hipHostMallocDefault);
    if (eh != hipSuccess) {
      hipFree (dev);
      return gt4hip_fail (ctx, GT4HIP_ENOMEM, "totals all-gather: pinned buffer: %s", hipGetErrorString (eh));
    }
    c->tot_dev = dev;
    c->tot_host = host;
  }
  for (uint32_t i = 0; i < n; i++) c->tot_host[i] = mine[i];
  HIPCHK (ctx, hipMemcpyAsync (c->tot_dev, c->tot_host, (size_t) n * 8, hipMemcpyHostToDevice, ctx->stream));
  const ncclResult_t e = r->AllGather (c->tot_dev, c->tot_dev + 8, n, ncclUint64, c->comm, ctx->stream);
  if (e != ncclSuccess) return gt4hip_fail (ctx, GT4HIP_ECOMM, "totals all-gather: %s", r->GetErrorString (e));
  HIPCHK (ctx, hipMemcpyAsync (c->tot_host + 8, c->tot_dev + 8, (size_t) n * c->n_ranks * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK (ctx, hipStreamSynchronize (ctx->stream));
  for (size_t i = 0; i < (size_t) n * c->n_ranks; i++) all[i] = c->tot_host[8 + i];
  return GT4HIP_OK;
}

extern "C" int gt4hip_comm_allgather_totals (gt4hip_comm *c, uint64_t n_words, uint64_t total_count, uint64_t *totals)
{
  const uint64_t mine[2] = { n_words, total_count };
  return gt4hip_comm_allgather_u64 (c, mine, 2, totals);
}

extern "C" int gt4hip_comm_gatherv (gt4hip_comm *c, const gt4hip_list *local, const uint64_t counts[], int root, gt4hip_list *gathered)
{
  if (!c || !counts || root < 0 || root >= c->n_ranks) return GT4HIP_EINVAL;
  gt4hip_context *ctx = c->ctx;
  const Rccl *r = rccl ();
  if (!r) return gt4hip_fail (ctx, GT4HIP_ECOMM, "%s", g_comm_err);
  const uint64_t mine = counts[c->rank];
  if (mine && (!local || local->n_words < mine)) return gt4hip_fail (ctx, GT4HIP_EINVAL, "gatherv: local list shorter than counts[rank]");
  HIPCHK (ctx, hipSetDevice (ctx->device));
  ncclResult_t e = ncclSuccess;
  if (c->rank != root) {
    if (mine) {
      e = r->GroupStart ();
      if (e == ncclSuccess) e = r->Send (local->dev, (size_t) mine * 3, ncclUint32, root, c->comm, ctx->stream);
      const ncclResult_t e2 = r->GroupEnd ();
      if (e == ncclSuccess) e = e2;
    }
  } else {
    uint64_t total = 0;
    for (int q = 0; q < c->n_ranks; q++) total += counts[q];
    if (!gathered || gathered->capacity < total) return gt4hip_fail (ctx, GT4HIP_EINVAL, "gatherv: root needs a list of capacity %llu", (unsigned long long) total);
    char *const base = (char *) gathered->dev;
    uint64_t off = 0;
    e = r->GroupStart ();
    for (int q = 0; q < c->n_ranks && e == ncclSuccess; q++) {
      if (q != root && counts[q]) e = r->Recv (base + off * GT4HIP_RECORD_BYTES, (size_t) counts[q] * 3, ncclUint32, q, c->comm, ctx->stream);
      off += counts[q];
    }
    const ncclResult_t e2 = r->GroupEnd ();
    if (e == ncclSuccess) e = e2;
    /* the root's own shard: a device-to-device copy on the same stream */
    if (e == ncclSuccess && mine) {
      uint64_t my_off = 0;
      for (int q = 0; q < root; q++) my_off += counts[q];
      if (base + my_off * GT4HIP_RECORD_BYTES != (char *) local->dev)
        HIPCHK (ctx, hipMemcpyAsync (base + my_off * GT4HIP_RECORD_BYTES, local->dev, (size_t) mine * GT4HIP_RECORD_BYTES, hipMemcpyDeviceToDevice, ctx->stream));
    }
    gathered->n_words = total;
  }
  if (e != ncclSuccess) return gt4hip_fail (ctx, GT4HIP_ECOMM, "gatherv: %s", r->GetErrorString (e));
  HIPCHK (ctx, hipStreamSynchronize (ctx->stream));
  return GT4HIP_OK;
}

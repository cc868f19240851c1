// fst.cpp - graph ingestion without OpenFST: a native restatement of what
// the reference's `simplefst` pybind module does on top of OpenFST 1.7.5
// (openfst_binding/src/fstext.cc): read a binary vector/standard FST from a file or from a byte
// offset inside a Kaldi ark, lay it out as the CSR tensors the loss consumes (FstToTensor,
// fstext.cc:19-117) and compute the leaky-HMM state prior (SetLeakyProbs, fstext.cc:120-171).
// Host-only, one-time per graph; exported through the same C ABI (include/pychain_hip.h).
//
// OpenFST itself is not part of the reference tree; the on-disk format read here is OpenFST's
// published binary layout for VectorFst<StdArc> (FstHeader + per-state {final weight, arc count,
// arcs{ilabel, olabel, weight, nextstate}}).  The reference holds no FST files: the reader is pinned by
// files assembled byte by byte from that layout (tests/golden/make_fst_bytes.py: plain file, header
// flags / unknown state count, two FSTs inside a Kaldi ark) and by write -> read round trips.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/pychain_hip.h"
#include "common.h"

namespace {

struct Arc { int32_t ilabel, olabel; float weight; int32_t nextstate; };
struct Fst {
  int64_t start = -1;
  std::vector<float> final_;              // tropical weights; +inf = not final
  std::vector<std::vector<Arc>> arcs;
  int64_t num_arcs() const { int64_t n = 0; for (auto& a : arcs) n += (int64_t)a.size(); return n; }
};

constexpr int32_t kFstMagic = 2125659606;

bool read_exact(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }
bool read_string(FILE* f, std::string* s) {
  int32_t n;
  if (!read_exact(f, &n, 4) || n < 0 || n > (1 << 20)) return false;
  s->resize((size_t)n);
  return n == 0 || read_exact(f, &(*s)[0], (size_t)n);
}

}  // namespace

extern "C" void* pychain_hip_fst_read(const char* filename, int64_t byte_offset) {
  if (!filename) { pychain_hip::fail(PYCHAIN_HIP_EINVAL, "fst_read: null filename"); return nullptr; }
  FILE* f = fopen(filename, "rb");
  if (!f) { pychain_hip::fail(PYCHAIN_HIP_EINVAL, "fst_read: cannot open %s", filename); return nullptr; }
  Fst* fst = new Fst;
  auto bail = [&](const char* why) -> void* {
    pychain_hip::fail(PYCHAIN_HIP_EINVAL, "fst_read: %s in %s", why, filename);
    fclose(f); delete fst; return nullptr;
  };
  if (byte_offset > 0 && fseek(f, (long)byte_offset, SEEK_SET) != 0) return bail("bad offset");   // ReadFstFromArk, fstext.cc:7-16
  int32_t magic, version, flags; uint64_t props; int64_t start, nstates, narcs;
  std::string ftype, atype;
  if (!read_exact(f, &magic, 4) || magic != kFstMagic) return bail("bad FST magic");
  if (!read_string(f, &ftype) || !read_string(f, &atype)) return bail("truncated header");
  if (ftype != "vector" || atype != "standard") return bail("only vector/standard FSTs are supported");
  if (!read_exact(f, &version, 4) || !read_exact(f, &flags, 4) || !read_exact(f, &props, 8) ||
      !read_exact(f, &start, 8) || !read_exact(f, &nstates, 8) || !read_exact(f, &narcs, 8))
    return bail("truncated header");
  if (flags & 3) return bail("embedded symbol tables are not supported");
  if (version < 2 || nstates < -1 || nstates > (int64_t)1 << 31) return bail("unsupported version or size");
  fst->start = start;
  // nstates == -1 (kNoStateId): the writer could not seek back to fill the count in; states follow until
  // the stream ends (VectorFstImpl::Read stops at the first final weight it cannot read)
  for (int64_t s = 0; nstates < 0 || s < nstates; s++) {
    float fw; int64_t na;
    if (!read_exact(f, &fw, 4)) { if (nstates < 0) break; return bail("truncated state"); }
    if (!read_exact(f, &na, 8) || na < 0 || na > (int64_t)1 << 31) return bail("truncated state");
    fst->final_.push_back(fw);
    fst->arcs.emplace_back((size_t)na);
    if (na && !read_exact(f, fst->arcs.back().data(), sizeof(Arc) * (size_t)na)) return bail("truncated arcs");
  }
  const int64_t ns = (int64_t)fst->final_.size();
  if (start < -1 || start >= ns) return bail("start state out of range");
  for (const auto& row : fst->arcs)
    for (const Arc& a : row)
      if (a.nextstate < 0 || a.nextstate >= ns) return bail("arc to a state out of range");
  fclose(f);
  return fst;
}

// Construction from arrays (what a caller without FST files, e.g. a graph compiler, would use).
extern "C" void* pychain_hip_fst_from_arcs(int32_t num_states, int32_t start, int64_t num_arcs,
                                            const int32_t* src, const int32_t* dst, const int32_t* ilabel,
                                            const float* weight, const float* final_weight) {
  if (num_states <= 0 || num_arcs < 0 || (num_arcs && (!src || !dst || !ilabel || !weight)) || !final_weight) {
    pychain_hip::fail(PYCHAIN_HIP_EINVAL, "fst_from_arcs: bad arguments");
    return nullptr;
  }
  Fst* fst = new Fst;
  fst->start = start;
  fst->final_.assign(final_weight, final_weight + num_states);
  fst->arcs.resize((size_t)num_states);
  for (int64_t k = 0; k < num_arcs; k++) {
    if (src[k] < 0 || src[k] >= num_states || dst[k] < 0 || dst[k] >= num_states) {
      pychain_hip::fail(PYCHAIN_HIP_EINVAL, "fst_from_arcs: arc %lld has a state out of range", (long long)k);
      delete fst; return nullptr;
    }
    fst->arcs[(size_t)src[k]].push_back(Arc{ilabel[k], ilabel[k], weight[k], dst[k]});
  }
  return fst;
}

extern "C" void pychain_hip_fst_free(void* h) { delete (Fst*)h; }
extern "C" int32_t pychain_hip_fst_num_states(const void* h) { return h ? (int32_t)((const Fst*)h)->final_.size() : -1; }
extern "C" int32_t pychain_hip_fst_start(const void* h) { return h ? (int32_t)((const Fst*)h)->start : -1; }
extern "C" int64_t pychain_hip_fst_num_arcs(const void* h) { return h ? ((const Fst*)h)->num_arcs() : -1; }

extern "C" int pychain_hip_fst_write(const void* h, const char* filename) {
  if (!h || !filename) return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "fst_write: null argument");
  const Fst* fst = (const Fst*)h;
  FILE* f = fopen(filename, "wb");
  if (!f) return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "fst_write: cannot open %s", filename);
  auto ws = [&](const char* s) { int32_t n = (int32_t)strlen(s); fwrite(&n, 4, 1, f); fwrite(s, 1, (size_t)n, f); };
  const int32_t version = 2, flags = 0; const uint64_t props = 0;
  const int64_t nstates = (int64_t)fst->final_.size(), narcs = fst->num_arcs();
  fwrite(&kFstMagic, 4, 1, f); ws("vector"); ws("standard");
  fwrite(&version, 4, 1, f); fwrite(&flags, 4, 1, f); fwrite(&props, 8, 1, f);
  fwrite(&fst->start, 8, 1, f); fwrite(&nstates, 8, 1, f); fwrite(&narcs, 8, 1, f);
  for (int64_t s = 0; s < nstates; s++) {
    const int64_t na = (int64_t)fst->arcs[(size_t)s].size();
    fwrite(&fst->final_[(size_t)s], 4, 1, f); fwrite(&na, 8, 1, f);
    if (na) fwrite(fst->arcs[(size_t)s].data(), sizeof(Arc), (size_t)na, f);
  }
  const bool ok = !ferror(f);
  fclose(f);
  return ok ? PYCHAIN_HIP_OK : pychain_hip::fail(PYCHAIN_HIP_EINVAL, "fst_write: I/O error on %s", filename);
}

// FstToTensor, fstext.cc:19-117.  Outputs are caller-allocated: [K,3] [K] [H,2] x2, [H].
// pdf = ilabel - 1 (:41), log_prob = -weight (:43-44), final = -Final(s) (:37); out-arcs per source in
// insertion order (:49-61), in-arcs per destination by ascending source then insertion order
// (:36-46,:63-76); exp() unless log_domain (:89-107).
extern "C" int pychain_hip_fst_to_tensors(const void* h, int log_domain,
                                          int32_t* fwd_trans, float* fwd_probs, int32_t* fwd_idx,
                                          int32_t* bwd_trans, float* bwd_probs, int32_t* bwd_idx, float* final_probs) {
  if (!h) return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "fst_to_tensors: null fst");
  const Fst* fst = (const Fst*)h;
  const int64_t H = (int64_t)fst->final_.size();
  const bool has_arcs = fst->num_arcs() > 0;    // an arc-less FST has zero-sized (possibly null) arc outputs
  if (!fwd_idx || !bwd_idx || !final_probs || (has_arcs && (!fwd_trans || !fwd_probs || !bwd_trans || !bwd_probs)))
    return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "fst_to_tensors: null argument");
  std::vector<int64_t> incount((size_t)H, 0);
  int64_t k = 0;
  for (int64_t s = 0; s < H; s++) {
    final_probs[s] = log_domain ? -fst->final_[(size_t)s] : expf(-fst->final_[(size_t)s]);
    fwd_idx[2 * s] = (int32_t)k;
    for (const Arc& a : fst->arcs[(size_t)s]) {
      if (a.ilabel < 1) return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "fst_to_tensors: epsilon input label on state %lld", (long long)s);
      fwd_trans[3 * k] = (int32_t)s; fwd_trans[3 * k + 1] = a.nextstate; fwd_trans[3 * k + 2] = a.ilabel - 1;
      fwd_probs[k] = log_domain ? -a.weight : expf(-a.weight);
      incount[(size_t)a.nextstate]++;
      k++;
    }
    fwd_idx[2 * s + 1] = (int32_t)k;
  }
  std::vector<int64_t> cursor((size_t)H);
  int64_t off = 0;
  for (int64_t s = 0; s < H; s++) { bwd_idx[2 * s] = (int32_t)off; cursor[(size_t)s] = off; off += incount[(size_t)s]; bwd_idx[2 * s + 1] = (int32_t)off; }
  for (int64_t s = 0; s < H; s++)          // ascending source, insertion order: stable bucket fill
    for (const Arc& a : fst->arcs[(size_t)s]) {
      const int64_t p = cursor[(size_t)a.nextstate]++;
      bwd_trans[3 * p] = (int32_t)s; bwd_trans[3 * p + 1] = a.nextstate; bwd_trans[3 * p + 2] = a.ilabel - 1;
      bwd_probs[p] = log_domain ? -a.weight : expf(-a.weight);
    }
  return PYCHAIN_HIP_OK;
}

// SetLeakyProbs, fstext.cc:120-171: 100 iterations of HMM propagation from the start state,
// per-state normalisation including the final weight, renormalised every step, averaged; float64.
extern "C" int pychain_hip_fst_leaky_probs(const void* h, float* out) {
  if (!h || !out) return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "fst_leaky_probs: null argument");
  const Fst* fst = (const Fst*)h;
  const size_t H = fst->final_.size();
  if (fst->start < 0 || (size_t)fst->start >= H) return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "fst_leaky_probs: no start state");
  const int num_iters = 100;
  std::vector<double> nf(H), cur(H, 0.0), nxt(H, 0.0), avg(H, 0.0);
  for (size_t s = 0; s < H; s++) {
    double tot = exp(-(double)fst->final_[s]);
    for (const Arc& a : fst->arcs[s]) tot += exp(-(double)a.weight);
    if (!(tot > 0.0)) return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "fst_leaky_probs: state %zu has no probability mass (fstext.cc:134 asserts)", s);
    nf[s] = 1.0 / tot;
  }
  cur[(size_t)fst->start] = 1.0;
  for (int it = 0; it < num_iters; it++) {
    for (size_t s = 0; s < H; s++) avg[s] += cur[s] * (1.0 / num_iters);
    for (size_t s = 0; s < H; s++) {
      const double p = cur[s] * nf[s];
      for (const Arc& a : fst->arcs[s]) nxt[(size_t)a.nextstate] += p * exp(-(double)a.weight);
    }
    double sum = 0.0;
    for (size_t s = 0; s < H; s++) sum += nxt[s];
    const double inv = 1.0 / sum;
    for (size_t s = 0; s < H; s++) { cur[s] = nxt[s] * inv; nxt[s] = 0.0; }
  }
  for (size_t s = 0; s < H; s++) out[s] = (float)avg[s];
  return PYCHAIN_HIP_OK;
}

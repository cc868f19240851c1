// api.hip - the C ABI of libpychain_hip.so (include/pychain_hip.h): argument checks,
// workspace carving, launches.  No allocation, no host synchronisation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

#include "../../include/pychain_hip.h"
#include "common.h"
#include "den_kernels.h"
#include "device_utils.h"
#include "num_kernels.h"
#include "plan_format.h"

namespace pychain_hip {
char* last_error_buffer() {
  static thread_local char buf[512] = "";
  return buf;
}
namespace {
// ---- settings: process-wide defaults + per-thread overrides, snapshotted per call (common.h:CallKnobs) ----------
std::mutex g_option_lock;
typedef std::map<std::string, std::string> OptionTable;
OptionTable& process_options() { static OptionTable t; return t; }
OptionTable& thread_options() { static thread_local OptionTable t; return t; }     // "" = unset for this thread
const char* const kKnownOptions[] = {"verbose", "den_phase_mask", "den_lazy", "den_dma", "den_segments", "den_pair", "gamma16",
                                     "debug_corrupt_row", "num_compat", "den_tseg", "den_tburn", "plan_split", "chain_slices", "den_sg", "den_cross", "den_q"};
bool known_option(const char* name) {
  if (!name) return false;
  for (const char* k : kKnownOptions) if (strcmp(k, name) == 0) return true;
  return false;
}
// effective value of one option as a COPY (the tables may change under another thread's set_option)
bool option_value(const char* name, std::string* out) {
  const OptionTable& th = thread_options();
  auto it = th.find(name);
  if (it != th.end()) { if (it->second.empty()) return false; *out = it->second; return true; }
  std::lock_guard<std::mutex> guard(g_option_lock);
  const OptionTable& pr = process_options();
  auto ip = pr.find(name);
  if (ip == pr.end()) return false;
  *out = ip->second;
  return true;
}
int option_int(const char* name, int dflt) {
  std::string v;
  return option_value(name, &v) ? atoi(v.c_str()) : dflt;
}
bool option_set(const char* name) { std::string v; return option_value(name, &v); }
size_t align256(size_t x) { return (x + 255) & ~size_t(255); }
int roundup64(int x) { return (x + 63) / 64 * 64; }
}  // namespace

CallKnobs call_knobs() {
  CallKnobs k;
  memset(&k, 0, sizeof(k));
  k.verbose = option_int("verbose", 0);
  k.den_phase_mask = option_int("den_phase_mask", 3) & 3;
  k.den_lazy = option_int("den_lazy", 1) ? 1 : 0;
  k.den_segments = option_int("den_segments", 0);
  k.gamma16 = option_set("gamma16");
  k.den_pair = option_int("den_pair", -1);
  k.den_dma = option_int("den_dma", -1);
  k.num_compat = option_int("num_compat", 0) ? 1 : 0;
  k.den_tseg = option_int("den_tseg", -1);
  k.den_tburn = option_int("den_tburn", 192);
  k.plan_split = option_int("plan_split", -1);
  k.chain_slices = option_int("chain_slices", -1);
  k.den_sg = option_int("den_sg", 1) ? 1 : 0;
  k.den_q = option_int("den_q", 0) ? 1 : 0;          // (off by default: the frame is bound by VALU issue, not by the LDS time it saves - DESIGN.md 3.16)
  k.den_cross = option_int("den_cross", 0) ? 1 : 0;   // (off by default: measured slower than the streamed occupancy launch - DESIGN.md 3.15)
  std::string v;
  if (option_value("debug_corrupt_row", &v)) {   // "den,b,t,scale" / "num,b,t,scale"
    char what[8] = ""; int b = 0, t = 0; float sc = 1.f;
    if (sscanf(v.c_str(), "%3[a-z],%d,%d,%f", what, &b, &t, &sc) == 4 && b >= 0 && t >= 0) {
      k.corrupt_what = strcmp(what, "den") == 0 ? 1 : (strcmp(what, "num") == 0 ? 2 : 0);
      k.corrupt_b = b; k.corrupt_t = t; k.corrupt_scale = sc;
    }
  }
  return k;
}
}  // namespace pychain_hip

using namespace pychain_hip;

extern "C" int pychain_hip_abi_version(void) { return PYCHAIN_HIP_ABI_VERSION; }
extern "C" const char* pychain_hip_last_error(void) { return last_error_buffer(); }
extern "C" int pychain_hip_set_option(const char* name, const char* value) {
  if (!known_option(name)) return fail(PYCHAIN_HIP_EINVAL, "set_option: unknown option '%s'", name ? name : "(null)");
  std::lock_guard<std::mutex> guard(g_option_lock);
  if (value && *value) process_options()[name] = value; else process_options().erase(name);
  return PYCHAIN_HIP_OK;
}
extern "C" int pychain_hip_set_thread_option(const char* name, const char* value) {
  if (!known_option(name)) return fail(PYCHAIN_HIP_EINVAL, "set_thread_option: unknown option '%s'", name ? name : "(null)");
  if (value) thread_options()[name] = value; else thread_options().erase(name);
  return PYCHAIN_HIP_OK;
}
extern "C" int pychain_hip_get_option(const char* name, char* buf, size_t buf_bytes) {
  if (!known_option(name)) return fail(PYCHAIN_HIP_EINVAL, "get_option: unknown option '%s'", name ? name : "(null)");
  std::string v;
  if (!option_value(name, &v)) return 0;
  if (buf && buf_bytes > 0) { strncpy(buf, v.c_str(), buf_bytes - 1); buf[buf_bytes - 1] = 0; }
  return (int)v.size();
}
static void set_process_int(const char* name, int v) {
  char s[32];
  snprintf(s, sizeof(s), "%d", v);
  pychain_hip_set_option(name, s);
}
extern "C" void pychain_hip_set_verbose_level(int level) { set_process_int("verbose", level); }
extern "C" int pychain_hip_get_verbose_level(void) { return call_knobs().verbose; }
extern "C" void pychain_hip_set_den_phase_mask(int mask) { set_process_int("den_phase_mask", mask & 3); }
extern "C" void pychain_hip_set_den_lazy(int on) { set_process_int("den_lazy", on ? 1 : 0); }

namespace {
bool den_call_is_pair(const DenArgs& a, int resident_slot_rows);
bool den_call_is_lazy(const DenArgs& a, int resident_slot_rows);
int den_call_shape(const DenArgs& a, int resident_slot_rows);
int device_cu_count();
}  // namespace
extern "C" int pychain_hip_den_kernel_names(int resident_slot_rows, int H, int D, int B, int plans_shared,
                                            char* buf, size_t buf_bytes) {
  if (!buf || buf_bytes == 0 || H <= 0 || D <= 0 || B <= 0) return fail(PYCHAIN_HIP_EINVAL, "den_kernel_names: bad arguments");
  DenArgs a;
  memset(&a, 0, sizeof(a));
  a.knobs = call_knobs();
  a.H = H; a.Hp = roundup64(H); a.D = D; a.B = B; a.frames_per_block = 32;
  a.plan_stride = (plans_shared & 1) ? 0 : 256;
  a.fused = (plans_shared & 2) ? 1 : 0;                  // (bit 1: the call is a fused loss)
  a.coef = 1e-5f;                                        // (the usual leaky-HMM coefficient: option den_q asks for one in [1e-8, 1])
  if (resident_slot_rows == PYCHAIN_HIP_HINT_GENERAL) {
    snprintf(buf, buf_bytes, "den_general_recursion_kernel,den_general_gamma_kernel");
    return PYCHAIN_HIP_OK;
  }
  a.pair = den_call_is_pair(a, resident_slot_rows) ? 1 : 0;
  a.lazy = den_call_is_lazy(a, resident_slot_rows) ? 1 : 0;
  a.shape = a.lazy ? den_call_shape(a, resident_slot_rows) : 0;
  a.sg = (a.lazy && a.shape == kShapeDma && den_sg_eligible(a, resident_slot_rows)) ? 1 : 0;
  // (the names of a call of the denominator alone that evaluates both launches: the crossing where option den_cross asks for it)
  { DenArgs long_call = a; long_call.T = 1 << 14;        // (of sequences long enough to have two halves)
    a.xf = (a.sg && !a.fused && 2 * a.B <= device_cu_count() && den_xf_eligible(long_call, resident_slot_rows)) ? den_xf_band() : 0; }
  snprintf(buf, buf_bytes, "%s,%s", den_recursion_kernel_name(a, resident_slot_rows),
           den_occupancy_kernel_name(a, (D + 63) / 64, resident_slot_rows));
  return PYCHAIN_HIP_OK;
}

extern "C" int pychain_hip_den_plan_info(const void* host_blob, size_t blob_bytes, int32_t info[8]) {
  if (!host_blob || !info || blob_bytes < sizeof(PlanHeader))
    return fail(PYCHAIN_HIP_EINVAL, "den_plan_info: null or truncated blob");
  if (((const int32_t*)host_blob)[0] == PLAN_MAGIC_GENERAL) {
    const GeneralPlanHeader* gh = (const GeneralPlanHeader*)host_blob;
    if (blob_bytes < sizeof(GeneralPlanHeader) || gh->version != PLAN_VERSION || (size_t)gh->total_bytes > blob_bytes ||
        (size_t)gh->total_bytes < sizeof(GeneralPlanHeader))
      return fail(PYCHAIN_HIP_EINVAL, "den_plan_info: not a plan of this library version");
    if ((int32_t)general_payload_hash(host_blob, (size_t)gh->total_bytes) != gh->payload_hash ||
        (int32_t)general_header_hash(*gh) != gh->reserved[0])
      return fail(PYCHAIN_HIP_EINVAL, "den_plan_info: plan does not match its checksums (corrupted or foreign file)");
    {
      const int64_t n = gh->total_bytes, H8 = (int64_t)gh->H * 8, K8 = (int64_t)gh->K * 8, K4 = (int64_t)gh->K * 4, Hp4 = (int64_t)gh->Hp * 4;
      const int64_t offs[12] = {gh->off_a_idx, gh->off_a_arc, gh->off_a_p, gh->off_b_idx, gh->off_b_arc, gh->off_b_p,
                                gh->off_g_idx, gh->off_g_arc, gh->off_g_p, gh->off_leaky, gh->off_init, gh->off_final};
      const int64_t lens[12] = {H8, K8, K4, H8, K8, K4, ((int64_t)gh->D + 1) * 4, K8, K4, Hp4, Hp4, Hp4};
      for (int i = 0; i < 12; i++)
        if (offs[i] < (int64_t)sizeof(GeneralPlanHeader) || offs[i] + lens[i] > n)
          return fail(PYCHAIN_HIP_EINVAL, "den_plan_info: a table of the plan lies outside the blob");
    }
    memset(info, 0, 8 * sizeof(int32_t));
    info[0] = gh->H; info[1] = gh->K; info[2] = gh->D;
    info[3] = (int32_t)(gh->total_bytes & 0x7fffffff);   // plan bytes = info[3] + (info[5] << 31): plans beyond 2 GiB keep their exact size
    info[5] = (int32_t)(gh->total_bytes >> 31);
    info[4] = PYCHAIN_HIP_HINT_GENERAL;          // launch hint: the general kernels (den_general.hip)
    return PYCHAIN_HIP_OK;
  }
  const PlanHeader* hd = (const PlanHeader*)host_blob;
  if (hd->magic != PLAN_MAGIC || hd->version != PLAN_VERSION || (size_t)hd->total_bytes > blob_bytes ||
      (size_t)hd->total_bytes < sizeof(PlanHeader))
    return fail(PYCHAIN_HIP_EINVAL, "den_plan_info: not a plan of this library version");
  // the kernels follow the blob's offsets, wave tables and packed LDS addresses unchecked: a blob that comes back from a
  // file (the on-disk plan cache) must be the bytes pychain_hip_den_plan_build wrote
  if ((int32_t)plan_payload_hash(host_blob, (size_t)hd->total_bytes) != hd->payload_hash ||
      (int32_t)plan_header_hash(*hd) != hd->header_hash)
    return fail(PYCHAIN_HIP_EINVAL, "den_plan_info: plan does not match its checksums (corrupted or foreign file)");
  if (!plan_header_in_bounds(*hd))
    return fail(PYCHAIN_HIP_EINVAL, "den_plan_info: a table of the plan lies outside the blob");
  memset(info, 0, 8 * sizeof(int32_t));
  info[0] = hd->H; info[1] = hd->K; info[2] = hd->D; info[3] = hd->total_bytes;
  int m = hd->alpha.max_wave_slot_rows;
  if (hd->beta.max_wave_slot_rows > m) m = hd->beta.max_wave_slot_rows;
  int gmm = hd->gamma.max_wave_slot_rows, gm2 = hd->gamma2.max_wave_slot_rows;
  if (m > 1023) m = 1023;
  if (gmm > 511) gmm = 511;                        // (9 bits since ABI 17: anything beyond the resident row counts means "stream the tail")
  if (gm2 > 1023) gm2 = 1023;
  // launch hint: recursion rows (10 bits) | occupancy rows, 16 waves (9 bits) << 10 | occupancy rows, 8 waves (7 bits) << 20
  if (gm2 > 127) gm2 = 127;                        // (7 bits since plan format 14; the two-frame kernel keeps at most 64 rows per wave)
  // bit 29: the plan holds the recursion tiles dealt to FOUR waves (small graphs: den_recursion_lazy_kernel<small>); the
  // recursion field is then the row count of THAT dealing (>= the 16-wave one: a kernel sized by it fits either)
  const bool small = hd->alpha4.nwaves == PLAN_REC4_WAVES && hd->beta4.nwaves == PLAN_REC4_WAVES &&
                     hd->rec4_max_wave_groups >= 1 && hd->rec4_max_wave_groups <= 4;
  if (small) m = std::max(m, std::max(hd->alpha4.max_wave_slot_rows, hd->beta4.max_wave_slot_rows));
  info[4] = m | (gmm << 10) | (gm2 << 20);
  if (small) info[4] |= 1 << 29;
  // bit 30: every recursion wave owns at most 4 groups (what den_recursion_lazy_kernel keeps in registers)
  if (hd->rec_max_wave_groups >= 1 && hd->rec_max_wave_groups <= 4) info[4] |= 1 << 30;
  // bit 28: a state sits on several positions of the beta numbering (plan.cpp, "states on several lanes"): not for
  // den_recursion_pair_kernel, whose normalise pass gives every position the constant c(t)
  if (hd->n_no_const > 0) info[4] |= 1 << 28;
  // bit 19: every state sits on ONE position of either numbering and every leaky probability is positive (>= 1e-12): the lazy
  // recursions' one-word state vectors (den_lazy.inc.h: MAP::kQ - alpha gathers a / cl, which needs cl > 0 everywhere)
  if (hd->n_no_const == 0 && hd->H == hd->graph_states) {
    const float* lk = (const float*)((const char*)host_blob + hd->off_leaky_a);
    bool pos = true;
    for (int i = 0; i < hd->H; i++) pos = pos && lk[i] >= 1e-12f && lk[i] <= 1e12f;
    if (pos) info[4] |= 1 << 19;
  }
  // bit 27: "pdf by state" - every arc entering a state carries one pdf: the lazy recursions' one-gather form (den_lazy.inc.h: SG)
  if (hd->flags & PLAN_FLAG_PDF_BY_STATE) info[4] |= 1 << 27;
  info[6] = hd->graph_states;                     // the graph's states (info[0]: positions of the longer side = what calls pass as num_states)
  info[7] = hd->H - hd->graph_states;            // positions added by states on several lanes (the longer side)
  return PYCHAIN_HIP_OK;
}

extern "C" size_t pychain_hip_den_workspace_min_bytes(int B, int T, int H, int D) {
  if (B <= 0 || T <= 0 || H <= 0 || D <= 0) return 0;
  const size_t Hp = roundup64(H);
  return align256(4 * (size_t)B * T * Hp) + align256(4 * (size_t)B * (T + 1) * Hp) +
         align256(8 * (size_t)B) + 256 + align256(36 * (size_t)B) /* progress counters (zeroed by every call) */ +
         2 * align256(4 * (size_t)B * (T + 2)) /* per-frame totals of the two recursions */ +
         align256(4 * (size_t)B * T) /* frame totals to check */ + align256(8 * (size_t)B) /* final dot products; den_finish_kernel's per-sequence side of the check */ + 256 +
         align256(4 * (size_t)B * 2 * kMaxTimeSegs * 2 * Hp) /* time segments: the speculated rows next to the segments (DenArgs::splice) */;
}
extern "C" size_t pychain_hip_den_workspace_bytes(int B, int T, int H, int D) {
  const size_t base = pychain_hip_den_workspace_min_bytes(B, T, H, D);
  return base ? base + align256(4 * (size_t)B * T * D) /* rows exp'd ahead of the recursions (DenArgs::ex) */ : 0;
}

namespace {
int32_t* tstate_of(const void* plans_dev);
int fill_den_args(DenArgs& a, const void* plans_dev, int64_t plan_stride_bytes, int hint, int H, int D,
                  const void* nnet_output, int x_dtype, int input_is_exp, const int64_t* seq_lengths,
                  int B, int T, float leaky_hmm_coefficient, float grad_scale,
                  float* objf_per_seq, void* grad, int32_t* bad_count,
                  void* workspace, size_t workspace_bytes, const char* who) {
  if (x_dtype < PYCHAIN_HIP_F32 || x_dtype > PYCHAIN_HIP_F16)
    return fail(PYCHAIN_HIP_EINVAL, "%s: unknown nnet_output_dtype %d", who, x_dtype);
  if (!plans_dev || !nnet_output || !seq_lengths || !objf_per_seq || !grad || !bad_count || !workspace)
    return fail(PYCHAIN_HIP_EINVAL, "%s: null pointer argument", who);
  if (B <= 0 || T <= 0 || H <= 0 || D <= 0)
    return fail(PYCHAIN_HIP_EINVAL, "%s: bad sizes B=%d T=%d H=%d D=%d", who, B, T, H, D);
  if ((H > 65535 || D > 65535) && hint != PYCHAIN_HIP_HINT_GENERAL)
    return fail(PYCHAIN_HIP_EUNSUPPORTED, "%s: more than 65535 states or pdfs need a plan in the general format (build it with "
                "pychain_hip_den_plan_build and pass the launch hint pychain_hip_den_plan_info reports)", who);
  // chain-computation.cc:68 asserts 0 < coefficient < 1 (compiled out under NDEBUG); here it is an error
  if (!(leaky_hmm_coefficient > 0.f && leaky_hmm_coefficient < 1.f))
    return fail(PYCHAIN_HIP_EINVAL, "%s: leaky_hmm_coefficient must be in (0,1), got %g", who,
                (double)leaky_hmm_coefficient);
  if (plan_stride_bytes < 0 || (plan_stride_bytes & 15))
    return fail(PYCHAIN_HIP_EINVAL, "%s: plan stride must be a non-negative multiple of 16", who);
  if (((uintptr_t)plans_dev | (uintptr_t)nnet_output | (uintptr_t)grad | (uintptr_t)workspace) & 15)
    return fail(PYCHAIN_HIP_EINVAL, "%s: plan, nnet_output, grad and workspace must be 16-byte aligned", who);
  if (workspace_bytes < pychain_hip_den_workspace_min_bytes(B, T, H, D))
    return fail(PYCHAIN_HIP_EWORKSPACE, "%s: workspace too small (%zu < %zu)", who, workspace_bytes,
                pychain_hip_den_workspace_min_bytes(B, T, H, D));
  memset(&a, 0, sizeof(a));
  a.plans = (const char*)plans_dev; a.plan_stride = plan_stride_bytes;
  a.x = (const float*)nnet_output; a.x_half = x_dtype; a.lengths = seq_lengths; a.objf = objf_per_seq; a.grad = (float*)grad; a.bad = bad_count;
  a.B = B; a.T = T; a.D = D; a.H = H; a.Hp = roundup64(H);
  a.input_is_exp = input_is_exp ? 1 : 0;
  a.frames_per_block = 32;      // measured at C3: 16 and 64 are both 1-3 % slower
  a.knobs = call_knobs();
  a.phase_mask = a.knobs.den_phase_mask;
  a.coef = leaky_hmm_coefficient; a.grad_scale = grad_scale;
  char* ws = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  a.alpha_store = (float*)ws;
  a.beta_store = (float*)(ws + align256(4 * (size_t)B * T * a.Hp));
  a.seq_progress = (int32_t*)(ws + align256(4 * (size_t)B * T * a.Hp) + align256(4 * (size_t)B * (T + 1) * a.Hp));   // [2][B]
  a.progress = (int32_t*)((char*)a.seq_progress + align256(8 * (size_t)B));
  a.stream_next = a.progress + 32;
  a.finish_count = a.progress + 40;
  a.occ_done = a.progress + 24; a.occ_done_target = 0;     // (a cache line of its own: polled while the queue head is drawn from)
  a.loss_out = nullptr; a.loss_num_objf = nullptr; a.loss_scale = 1.f; a.loss_norm_dev = nullptr; a.bad_words = 1;
  a.stream = 0;
  a.xprog = (int32_t*)((char*)a.progress + 256);           // [2][B][kExMaxQ], then xnan [B]
  a.xnan = a.xprog + 2 * (size_t)B * kExMaxQ;
  a.ex_nr = 0; a.ex_q = 0;
  a.use_ex = 0;
  a.tot_a = (float*)((char*)a.xprog + align256(36 * (size_t)B));
  a.tot_b = (float*)((char*)a.tot_a + align256(4 * (size_t)B * (T + 2)));
  a.gtot = (float*)((char*)a.tot_b + align256(4 * (size_t)B * (T + 2)));
  a.fin_dot = (float*)((char*)a.gtot + align256(4 * (size_t)B * T));
  // (behind everything else, and only in a workspace of the full size: DenArgs::ex)
  a.splice = (float*)((char*)a.fin_dot + align256(8 * (size_t)B) + 256);
  a.redo = a.progress + 48; a.redo_if = 0; a.tseg = 0; a.tburn = 0;
  // (the plan's burn-in controller, unless the caller pins the cut or the burn-in with an option of its own)
  a.tstate = (option_set("den_tseg") || option_set("den_tburn")) ? nullptr : tstate_of(plans_dev);
  a.ex = workspace_bytes >= pychain_hip_den_workspace_bytes(B, T, H, D) ? (float*)((char*)a.splice + align256(4 * (size_t)B * 2 * kMaxTimeSegs * 2 * a.Hp)) : nullptr;
  a.lazy = 0;
  a.check = 0; a.check_all = a.knobs.verbose >= 1 ? 1 : 0;
  a.sig_n = 0;
  a.gam_seg = 0; a.gam_nseg = 0;
  return PYCHAIN_HIP_OK;
}
}  // namespace

// ---- library-owned side streams (the only hidden state besides the option table): one for the numerator
// recursion, one for the occupancy launches that overlap the denominator recursion ------------------
namespace {
constexpr int kMaxSegments = 16;
struct SideStream {
  hipStream_t stream = nullptr, stream2 = nullptr;
  hipEvent_t fork = nullptr, join = nullptr, seg[kMaxSegments] = {}, join2 = nullptr;
  bool ready = false;       // every stream and event below was created
};
// One set per (device, caller's stream): two calls in flight on two streams of one device must not share
// the side streams' events (a second hipEventRecord would move the event the first call still waits for),
// and calls on ONE stream are ordered, so they may.  Entries live for the life of the process.
SideStream* side_streams_for(hipStream_t caller) {
  static std::map<std::pair<int, hipStream_t>, SideStream> table;
  static std::mutex create_lock;              // first use may come from several host threads
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> guard(create_lock);
  SideStream& s = table[std::make_pair(dev, caller)];
  if (!s.ready) {
    if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
    // stream2: den_exp_rows_kernel, whose rows the recursions wait for (dispatched first: highest priority), then the gate(s)
    // and the occupancy launches that overlap the recursions
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) greatest = 0;
    if (hipStreamCreateWithPriority(&s.stream2, hipStreamNonBlocking, greatest) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&s.fork, hipEventDisableTiming | hipEventReleaseToDevice) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&s.join, hipEventDisableTiming | hipEventReleaseToDevice) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&s.join2, hipEventDisableTiming | hipEventReleaseToDevice) != hipSuccess) return nullptr;
    for (int i = 0; i < kMaxSegments; i++)
      if (hipEventCreateWithFlags(&s.seg[i], hipEventDisableTiming | hipEventReleaseToDevice) != hipSuccess) return nullptr;
    s.ready = true;
  }
  return &s;
}

// Number of time segments the denominator is cut into so that the occupancy pass of the frames
// whose alpha'/beta rows already exist runs on idle CUs WHILE the recursions continue
// (2B persistent workgroups leave the other CUs free).  1 = no overlap.
int den_segments(const DenArgs& a) {
  const int T = a.T;
  if (a.knobs.den_segments >= 1 && a.knobs.den_segments <= kMaxSegments) return a.knobs.den_segments;
  // What limits the overlap is CU time: after T/2 the occupancy pass has the ~128 idle CUs only (less the
  // numerator's), on which its ~1 ms of whole-chip work takes longer than the rest of the recursion, so
  // the last launch (the frames that only become computable at the very end) is exposed.  Each further
  // segment halves that launch but costs a launch (gate + kernel, ~0.05 ms of side-stream time).
  // Measured at C3 (T=1500), whole step, gated schedule with compact grids:
  //   3 -> 3.85 ms, 4 -> 3.74, 5 -> 3.81, 6 -> 3.90   (C4, T=2000: 3 -> 6.86, 4 -> 6.84, 5 -> 7.03)
  // (history: with one recursion launch per segment and full occupancy grids it was 3 -> 4.43, 4 -> 4.52)
  // shorter batches (B=64, same graph; ms per step for 1 / 2 / 3 / 4 segments): T=896: 2.79 / 2.71 / 2.42 / 2.37,
  // T=640: 2.01 / 1.93 / 1.72 / 1.72, T=384: 1.21 / 1.22 / 1.07 / 1.11
  if (T >= 768) return 4;
  if (T >= 256) return 3;
  // small graphs in four-wave workgroups leave most of the chip idle and their frames are short: the occupancy pass
  // overlaps from 64 frames on (C2, T = 150)
  if (a.lazy && a.shape == kShapeSmall && T >= 64) return 2;
  return 1;
}

// Which form the stored rows of this call have (decided from the same inputs by the forward call and by a
// later chain_loss_backward on its workspace).
// Two sequences per recursion workgroup (den_pair.inc.h) pay once the 2B one-sequence workgroups would fill the
// chip: the recursions then run on half of it and the occupancy launches and the numerator on the other half.
// Option den_pair: "1" wherever the shape allows (the tests), "0" never.
int device_cu_count() {
  static std::mutex lock;
  static std::map<int, int> cus;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::lock_guard<std::mutex> guard(lock);
  auto it = cus.find(dev);
  if (it != cus.end()) return it->second;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  return cus[dev] = n;
}
bool den_call_is_small(const DenArgs& a, int resident_slot_rows) {
  return a.knobs.den_dma != 0 && den_small_eligible(a, resident_slot_rows);
}
bool den_call_is_pair(const DenArgs& a, int resident_slot_rows) {
  if (!den_pair_eligible(a, resident_slot_rows)) return false;
  if (a.knobs.den_pair >= 0) return a.knobs.den_pair != 0;
  // small graphs run in four-wave workgroups, several to a CU: nothing to gain from pairing sequences
  if (a.knobs.den_lazy && den_call_is_small(a, resident_slot_rows)) return false;
  // measured on the C3 graph (round 5, 256 CUs).  Fused step: B = 96 one-sequence workgroups 4.40 ms, pairs 4.58; B = 104:
  // 5.36 / 4.58; B = 128: 6.60 / 5.24 - below about 100 sequences the one-sequence workgroups leave enough of the chip to the
  // occupancy launches and the numerator, and their chain is shorter.  The denominator ALONE has the chip to itself: one-
  // sequence workgroups while they all fit (B = 104: 3.77 against 4.63 with pairs, 112: 3.95 / 4.62, 128: 4.36 / 4.80)
  if (a.fused) return 64 * a.B >= 25 * device_cu_count();
  return 2 * a.B > device_cu_count();
}
// the nnet-output rows of the lazy recursions come in by LDS-direct loads (default wherever a lazy shape fits: on the
// 16-wave map of C3 it is 2 % faster than rows through registers and bit-identical, and it is what makes rows of
// 4096 < D <= 9216 pdfs - C4 - fit a 128-VGPR wave at all); option den_dma = "0": rows through registers, i.e. the
// 16-wave map for D <= 4096 and den_recursion_kernel beyond
bool den_call_is_dma(const DenArgs& a, int resident_slot_rows) {
  return a.knobs.den_dma != 0 && den_dma_eligible(a, resident_slot_rows);
}
int den_call_shape(const DenArgs& a, int resident_slot_rows) {     // DenArgs::shape
  if (den_call_is_small(a, resident_slot_rows)) return kShapeSmall;
  return den_call_is_dma(a, resident_slot_rows) ? kShapeDma : kShapeRegs;
}
bool den_call_is_lazy(const DenArgs& a, int resident_slot_rows) {
  return a.knobs.den_lazy && !den_call_is_pair(a, resident_slot_rows) &&
         (den_lazy_eligible(a, resident_slot_rows) || den_call_is_small(a, resident_slot_rows) || den_call_is_dma(a, resident_slot_rows));
}

// Time segments (DenArgs::tseg): how many a call's (sequence, direction) recursions are cut into.  With few
// sequences the chain of T dependent frames IS the step and most CUs idle; S segments started `burn` frames outside
// themselves run T / S + burn frames each on 2 B S workgroups.  The price: the burn-in frames (CU-time) and the occupancy launch no
// longer overlapping the recursions (a frame's rows come from four workgroups, not two).  Chosen where the estimate says it
// pays by more than 5 %: per frame ~1.94 us (2.2 us for rows beyond 4096 pdfs), occupancy ~3.5 ps per frame and pdf of whole-chip
// time; the grid must leave every workgroup a CU (a fused call: half of the chip stays with the numerator).  The fused loss at
// B = 64 is never cut (CU-time-bound: DESIGN.md §4); the denominator alone at B = 64 runs two segments, C4 (B = 32) four.
int den_time_segments(const DenArgs& a, bool fused) {
  const int want = a.knobs.den_tseg;
  // (not a function of the verbose level: a debug run computes what production computes - ADVICE r5; the per-frame check of
  // verbose >= 1 reads the totals and occupancy sums a cut call stores like an uncut one)
  if (want == 0 || want == 1 || !a.lazy || a.shape != kShapeDma) return 1;
  const int burn = a.knobs.den_tburn;
  if (burn < 1) return 1;
  // (a fused call: a quarter of the chip stays with the numerator - measured on 256 CUs, the C3 graph: B = 24 in 4 segments
  // (192 workgroups) 1.58 ms against 2.25 in 2; B = 40 / 48 in 2 (160 / 192): 2.53 / 2.67 against 3.02 uncut; B = 32 in 4 (256): 2.59
  // against 2.41 in 2; B = 56 in 2 (224): as uncut)
  const int cus = fused ? device_cu_count() * 3 / 4 : device_cu_count();
  int best = 1;
  const double f = a.D > 4096 ? 2.2e-6 : 1.94e-6, occ = 3.5e-12 * (double)a.D * (double)a.B * (double)a.T;
  // (not cut: the chain, 5 % of head and tail, and the part of the streamed occupancy launch that is left when the recursions
  // end - about a third of it with sequences of one length; cut: the occupancy launch follows the recursions, + three launches)
  double best_t = 0.95 * (1.05 * (double)a.T * f + (fused ? 0.1 : 0.35) * occ);
  for (int S = 2; S <= kMaxTimeSegs; S *= 2) {
    // (a forced count only has to fit the chip: the segments wait for nobody)
    if (2 * a.B * S > (want == S ? device_cu_count() : cus) || a.T < 2 * burn) continue;
    if (want == S) return S;
    const double t = ((double)a.T / S + burn) * f + occ + 6e-5;
    if (want < 0 && t < best_t) { best = S; best_t = t; }
  }
  return best;
}

// Would a call of the denominator alone, given a workspace with the [B,T,D] buffer, exp its rows ahead of the recursions (§3.9)?
// (a.lazy / a.shape / a.pair decided.)  The recursion workgroups spin on rows that launch writes and each takes a whole CU, so the
// launch must find CUs of its own whatever the order of dispatch: at least a quarter of the chip stays free of recursion
// workgroups, else the recursions exp their rows themselves (2B >= the CU count with pairing off could hang).
bool den_would_exp_rows_ahead(const DenArgs& a) {
  // (not where the call is cut into time segments: the rows are written from the sequence ends inwards, a segment starts inside;
  // not for the one-gather form of a "pdf by state" plan: its beta recursion reads its rows a frame ahead of the others)
  // (round 6: NOT for the 16-wave maps any more - since the rows' clamp / exp sits late in the arc phase and the frame lost a dozen
  // instructions, the recursions with their own rows are the faster ones in the only calls that still took them, the uncut ones of
  // 65 .. 96 sequences: C3 graph, denominator alone, B = 80: 3.35 -> 3.23 ms, B = 96: 4.0 -> 3.6 ms; C4's graph and rows, B = 80,
  // T <= 1000: 4.25 -> 3.25 ms (profiles/r06_rows_ahead.txt); four-wave workgroups keep them (C2: 0.208 against 0.214 ms); option
  // den_dma = 3: wherever the shape allows)
  const bool pays = a.shape == kShapeSmall || a.knobs.den_dma == 3;
  return a.lazy && !a.sg && (a.shape == kShapeDma || a.shape == kShapeSmall) && a.knobs.den_dma != 2 && pays && !a.input_is_exp &&
         a.D % 4 == 0 && a.D <= 4 * 5 * 512 && a.T >= 64 && 4 * den_recursion_blocks(a) <= 3 * device_cu_count() &&
         den_time_segments(a, false) == 1;
}

// 2-byte network outputs (DenArgs::x_half) are read as they are - and the gradient written in the same type - by the lazy
// recursions with LDS-direct rows, the pair recursion and both occupancy kernels in their float4-chunk forms; rows of a
// multiple of 8 pdfs (a lane's 16 raw bytes are 8 elements: all inside the row or all past it).  Every other kernel family
// (general plans, the two-barrier recursion, rows through registers, per-sequence plans of the gated schedule included)
// reads fp32: the caller up-casts (pychain_amd/native.py does).
bool den_call_half_native(const DenArgs& a0, int hint) {
  if (hint == PYCHAIN_HIP_HINT_GENERAL || a0.D % 8 != 0) return false;
  DenArgs a = a0;
  a.lazy = den_call_is_lazy(a, hint) ? 1 : 0;
  a.shape = a.lazy ? den_call_shape(a, hint) : 0;
  a.pair = den_call_is_pair(a, hint) ? 1 : 0;
  const bool rec_ok = a.pair || (a.lazy && (a.shape == kShapeDma || a.shape == kShapeSmall));
  return rec_ok && den_occupancy_half_ok(a, (a.D + 63) / 64, hint);
}
// the numerator's tile recursions stage 2-byte rows in their float4-chunk forms (num_fb_kernel); the general and the
// reference-arithmetic kernels read fp32
bool num_half_native(const NumArgs& a) {
  return !a.general && !a.compat && a.D % 4 == 0 && a.D <= 4 * 8 * 512;
}

// option debug_corrupt_row: row[0..n) *= scale, between the recursion and the occupancy launches
__global__ void scale_row_kernel(float* row, int n, float scale) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) row[i] *= scale;
}
hipError_t launch_scale_row(float* row, int n, float scale, hipStream_t st) {
  hipLaunchKernelGGL(scale_row_kernel, dim3(1), dim3(256), 0, st, row, n, scale);
  return hipGetLastError();
}

// Every word a call's launches count in - the caller's bad count(s) and the workspace's counters (per-sequence progress, gate
// counters, the queue head of the streamed occupancy pass, den_finish_kernel's arrival counter) - zeroed by ONE small launch
// (two hipMemsetAsync are two fill kernels with a gap between them: 18 us at the head of every step).
// (`norm` / `inv_out`: the fused loss with a device-side normaliser - *inv_out = 1 / *norm for the occupancy launches of the
// call, which scale the gradient by it; inv_out lies inside p1's range and is written behind its zeroing)
__global__ void zero_words_kernel(int32_t* p0, int n0, int32_t* p1, int n1, const float* norm, float* inv_out) {
  for (int i = threadIdx.x; i < n0; i += blockDim.x) p0[i] = 0;
  for (int i = threadIdx.x; i < n1; i += blockDim.x) p1[i] = 0;
  if (inv_out) {
    __syncthreads();
    if (threadIdx.x == 0) *inv_out = 1.0f / *norm;
  }
}
hipError_t launch_zero_words(int32_t* p0, int n0, int32_t* p1, int n1, hipStream_t st, const float* norm = nullptr, float* inv_out = nullptr) {
  hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(256), 0, st, p0, n0, p1, n1, norm, inv_out);
  return hipGetLastError();
}
int den_counter_words(const DenArgs& a) { return (int)((align256(8 * (size_t)a.B) + 256 + align256(36 * (size_t)a.B)) / 4); }

// recursion + occupancy launches of one denominator call; `occupancy` = false: recursion only
// `gamma_wait`: event every occupancy launch has to wait for (the numerator rows it folds in), or null
// `zeroed`: an event the caller recorded on `st` behind the zeroing of the counters (the fused loss forks its numerator
// stream there), or null: recorded here.  An event record costs the caller's stream ~7 us in front of the recursion launch.
// `finish_early`: in/out - in: den_finish_kernel may be launched here (the caller has nothing else it must wait for);
// out: it was (behind the recursion launch of the streamed schedule, DenArgs::occ_done), the caller must not launch it again
hipError_t run_den_launches(DenArgs& a, int resident_slot_rows, bool occupancy, hipStream_t st, const char** why,
                            hipEvent_t gamma_wait, hipEvent_t zeroed = nullptr, bool* finish_early = nullptr) {
  const bool may_finish = finish_early && *finish_early;
  if (finish_early) *finish_early = false;
  const int gmax = (a.D + 63) / 64;
  const int user_mask = a.phase_mask;
  // (the counters the launches of a call share were zeroed with the bad count by the caller: launch_zero_words)
  if (resident_slot_rows == PYCHAIN_HIP_HINT_GENERAL) {
    // a plan in the general format: den_general.hip, no overlap (rows in den_recursion_kernel's normalised form)
    a.lazy = 0; a.pair = 0; a.shape = 0;
    a.check = (occupancy && user_mask == 3) ? 1 : 0;
    const bool corrupt_g = a.knobs.corrupt_what == 1 && a.knobs.corrupt_b < a.B && a.knobs.corrupt_t < a.T;
    a.phase_mask = user_mask & 1;
    hipError_t eg = a.phase_mask ? launch_den_general(a, st) : hipSuccess;
    if (eg == hipSuccess && corrupt_g && a.phase_mask)
      eg = launch_scale_row(a.alpha_store + ((size_t)a.knobs.corrupt_b * a.T + a.knobs.corrupt_t) * a.Hp, a.Hp, a.knobs.corrupt_scale, st);
    if (eg == hipSuccess && gamma_wait) eg = hipStreamWaitEvent(st, gamma_wait, 0);
    a.phase_mask = occupancy ? (user_mask & 2) : 0;
    if (eg == hipSuccess && a.phase_mask) eg = launch_den_general(a, st);
    a.phase_mask = user_mask;
    return eg;
  }
  // (a corrupted row - option debug_corrupt_row - is written between the recursion and the occupancy launches: no overlap)
  const bool corrupt = a.knobs.corrupt_what == 1 && a.knobs.corrupt_b < a.B && a.knobs.corrupt_t < a.T;
  a.lazy = den_call_is_lazy(a, resident_slot_rows) ? 1 : 0;
  a.shape = a.lazy ? den_call_shape(a, resident_slot_rows) : 0;
  // two sequences per workgroup once the 2B one-sequence workgroups would fill the chip (option den_pair: 1 always
  // where the shape allows, 0 never); rows in den_recursion_kernel's form
  a.pair = den_call_is_pair(a, resident_slot_rows) ? 1 : 0;
  a.sg = (a.lazy && a.shape == kShapeDma && den_sg_eligible(a, resident_slot_rows)) ? 1 : 0;   // (a "pdf by state" plan in the one-gather form)
  // ... whose recursions emit the occupancies of their own second halves themselves (DenArgs::xf): a call of the denominator alone
  // that evaluates both launches; the occupancy launch then follows the recursions and handles the bands around the middles
  // (each direction waits for rows of the other: every workgroup of the launch must be resident at once)
  a.xf = (a.sg && occupancy && user_mask == 3 && !corrupt && zeroed == nullptr && 2 * a.B <= device_cu_count() &&
          den_xf_eligible(a, resident_slot_rows)) ? den_xf_band() : 0;
  const int nseg = (occupancy && user_mask == 3 && !a.check_all && !corrupt && !a.xf) ? den_segments(a) : 1;
  // the invariant check (DenArgs::tot_a) needs the recursions and the occupancy launches of ONE call
  a.check = (occupancy && user_mask == 3) ? 1 : 0;
  hipError_t e = hipSuccess;
  // Rows exp'd ahead of the lazy recursions by a launch of its own on the side stream (DenArgs::ex), whichever schedule
  // follows (option den_dma = 2: the recursions clamp / exp their rows themselves, as they do for short sequences, rows
  // that are not a multiple of four pdfs and callers that hand in exp'd rows) - in a call of the denominator alone and a
  // workspace that holds the buffer.  NOT in the fused loss (`zeroed`): there the CUs the recursions leave idle are booked - the
  // numerator until T/2, the occupancy launch from there to the end - and the launch's registers and HBM traffic held the
  // numerator back by 1 ms (C3: 3.02 -> 3.31 ms per step with it, although the recursion itself ran 4 % faster:
  // profiles/r04_u_C3_step_timeline_rows_exp_ahead_in_the_fused_step.txt).
  const bool exp_ahead = a.ex != nullptr && zeroed == nullptr && den_would_exp_rows_ahead(a) && (user_mask & 1) != 0;
  SideStream* const side_pre = exp_ahead ? side_streams_for(st) : nullptr;
  if (exp_ahead && !side_pre) { *why = "cannot create the side streams"; return hipErrorInvalidValue; }
  bool forked = false;                                      // stream2 waits for the zeroed counters (and the caller's x)
  auto fork_stream2 = [&](SideStream* side) -> hipError_t {
    if (forked) return hipSuccess;
    forked = true;
    hipError_t ef = zeroed ? hipSuccess : hipEventRecord(side->seg[0], st);
    if (ef == hipSuccess) ef = hipStreamWaitEvent(side->stream2, zeroed ? zeroed : side->seg[0], 0);
    return ef;
  };
  if (exp_ahead) {
    e = fork_stream2(side_pre);
    den_exp_rows_shape(a, device_cu_count(), &a.ex_nr, &a.ex_q);
    if (e == hipSuccess) e = launch_den_exp_rows(a, side_pre->stream2);
    a.use_ex = 1;
  }
  // Time segments: recursion launch over 2 B S workgroups, the check of every speculated row, the ordinary recursion launch as
  // a fallback that runs only if a row did not verify, then the occupancy launch (no overlap: den_time_segments)
  const int tseg = exp_ahead ? 1 : den_time_segments(a, zeroed != nullptr);   // (den_would_exp_rows_ahead: false where this says > 1)
  if (tseg > 1) {
    const int mask = occupancy ? user_mask : (user_mask & 1);
    if (mask & 1) {
      a.phase_mask = 1; a.tseg = tseg; a.tburn = a.knobs.den_tburn;
      e = launch_den(a, gmax, resident_slot_rows, st, why);
      if (e == hipSuccess) e = launch_den_splice_check(a, st);
      const int keep = a.tseg;
      a.tseg = 0; a.redo_if = 1;
      if (e == hipSuccess) e = launch_den(a, gmax, resident_slot_rows, st, why);
      a.redo_if = 0; a.tseg = keep;                       // (den_finish_kernel: an inner segment's NaN report)
      if (e == hipSuccess && corrupt)
        e = launch_scale_row(a.alpha_store + ((size_t)a.knobs.corrupt_b * a.T + a.knobs.corrupt_t) * a.Hp, a.Hp, a.knobs.corrupt_scale, st);
    }
    if (e == hipSuccess && gamma_wait && (mask & 2)) e = hipStreamWaitEvent(st, gamma_wait, 0);
    if (e == hipSuccess && (mask & 2)) { a.phase_mask = 2; e = launch_den(a, gmax, resident_slot_rows, st, why); }
    a.phase_mask = user_mask;
    return e;
  }
  if (nseg <= 1) {
    const int mask = occupancy ? user_mask : (user_mask & 1);
    if ((gamma_wait || corrupt) && (mask & 2)) {
      a.phase_mask = mask & 1;
      if (a.phase_mask) e = launch_den(a, gmax, resident_slot_rows, st, why);
      if (e == hipSuccess && corrupt && a.phase_mask)
        e = launch_scale_row(a.alpha_store + ((size_t)a.knobs.corrupt_b * a.T + a.knobs.corrupt_t) * a.Hp, a.Hp, a.knobs.corrupt_scale, st);
      if (e == hipSuccess && gamma_wait) e = hipStreamWaitEvent(st, gamma_wait, 0);
      a.phase_mask = 2;
      if (e == hipSuccess) e = launch_den(a, gmax, resident_slot_rows, st, why);
    } else {
      a.phase_mask = mask;
      e = launch_den(a, gmax, resident_slot_rows, st, why);
    }
    a.phase_mask = user_mask;
    if (exp_ahead) {                                        // (long done: the recursions have read every row of it)
      a.use_ex = 0;
      if (e == hipSuccess) e = hipEventRecord(side_pre->join2, side_pre->stream2);
      if (e == hipSuccess) e = hipStreamWaitEvent(st, side_pre->join2, 0);
    }
    return e;
  }
  SideStream* side = side_streams_for(st);
  if (!side) { *why = "cannot create the side streams"; return hipErrorInvalidValue; }
  // Frame t becomes computable after max(t, L-1-t) recursion steps, i.e. nothing before T/2 and
  // then ever faster: segment ends at T/2, 3T/4, 7T/8, ... so every occupancy launch but the
  // last overlaps the next recursion segment and the last one holds ~2^-(nseg-1) of the frames.
  for (int s = 0; s < nseg; s++) {
    const double frac = s == nseg - 1 ? 1.0 : 1.0 - 1.0 / (double)(2 << s);
    a.seg_bound[s] = s == nseg - 1 ? a.T : ((int)(frac * a.T) + 31) / 32 * 32;
  }
  if (a.knobs.den_segments == 0 && den_stream_eligible(a, gmax, resident_slot_rows)) {
    // Streamed schedule (DenArgs::stream, den_kernels.hip: stream_take): ONE recursion launch whose workgroups report
    // per-sequence progress, ONE persistent occupancy launch on the side stream - released when every recursion
    // workgroup has passed T/2 (nothing is computable before; the numerator has the idle CUs until then) - that draws
    // rings of frames from a queue in the order in which they become computable.  No segment granularity, no exposed last
    // launch: what is left when the recursions end is the last ring of every sequence.
    a.stream = 1; a.sig_n = 1; a.stream_blocks = device_cu_count();
    a.seg_bound[0] = std::min(a.T, (a.T / 2 + 31) / 32 * 32);
    // (The other way round - the recursion launch on the side stream, gate and occupancy launch on the caller's, so that the
    // launch that ends last is followed in queue order - was measured: the wake-up of a queue that waits for another queue's
    // event costs 11-19 us wherever it sits, and there the recursion paid it at the head of the step:
    // profiles/r04_i_C3_step_timeline_recursion_on_side_stream.txt.)
    if (e == hipSuccess) e = fork_stream2(side);              // (the occupancy launch must see the zeroed counters)
    a.phase_mask = 1;
    if (e == hipSuccess) e = launch_den(a, gmax, resident_slot_rows, st, why);
    if (e == hipSuccess) e = launch_den_gate(a.progress, den_recursion_blocks(a), a.bad, side->stream2);
    if (e == hipSuccess && gamma_wait) e = hipStreamWaitEvent(side->stream2, gamma_wait, 0);
    a.phase_mask = 2; a.gam_nseg = 0; a.gam_seg = 0; a.stream = 3;
    a.occ_done_target = may_finish ? a.stream_blocks : 0;
    if (e == hipSuccess) e = launch_den(a, gmax, resident_slot_rows, side->stream2, why);
    if (e == hipSuccess) e = hipEventRecord(side->join2, side->stream2);
    a.phase_mask = user_mask; a.sig_n = 0; a.stream = 0; a.use_ex = 0;
    if (e == hipSuccess && may_finish) { e = launch_den_finish(a, st); *finish_early = true; }   // (DenArgs::occ_done)
    a.occ_done_target = 0;
    if (e == hipSuccess) e = hipStreamWaitEvent(st, side->join2, 0);
    return e;
  }
  // Gated schedule (rounds 1-2; the fallback for the two-barrier recursion and per-sequence plans, and what option
  // den_segments = n asks for): ONE recursion launch; its workgroups count themselves into progress[s] when their steps
  // below seg_bound[s] are done, and a one-wave gate kernel in front of occupancy launch s (side stream) waits for all
  // of them.  (One recursion launch per segment with stream events in between - the first form of this schedule - measured
  // 2 % slower and is gone: profiles/r01_*.)
  a.sig_n = nseg - 1;
  if (e == hipSuccess) e = fork_stream2(side);                // the gates must see the zeroed counters
  a.phase_mask = 1;
  if (e == hipSuccess) e = launch_den(a, gmax, resident_slot_rows, st, why);
  a.phase_mask = 2; a.gam_nseg = nseg;
  for (int s = 0; s < nseg - 1 && e == hipSuccess; s++) {
    a.gam_seg = s;
    e = launch_den_gate(a.progress + s, den_recursion_blocks(a), a.bad, side->stream2);
    if (e == hipSuccess && s == 0 && gamma_wait) e = hipStreamWaitEvent(side->stream2, gamma_wait, 0);
    if (e == hipSuccess) e = launch_den(a, gmax, resident_slot_rows, side->stream2, why);
  }
  // the last occupancy launch follows the recursion in stream order on the caller's stream
  a.gam_seg = nseg - 1;
  if (e == hipSuccess && gamma_wait) e = hipStreamWaitEvent(st, gamma_wait, 0);
  if (e == hipSuccess) e = launch_den(a, gmax, resident_slot_rows, st, why);
  if (e == hipSuccess) e = hipEventRecord(side->join2, side->stream2);
  if (e == hipSuccess) e = hipStreamWaitEvent(st, side->join2, 0);
  a.phase_mask = user_mask; a.gam_nseg = 0; a.sig_n = 0; a.use_ex = 0;
  return e;
}
// ... and, behind all of them on the caller's stream, den_finish_kernel: objf from the per-frame totals and
// the reference's invariant check (DenArgs::tot_a)
hipError_t run_den(DenArgs& a, int resident_slot_rows, bool occupancy, hipStream_t st, const char** why,
                   hipEvent_t gamma_wait = nullptr) {
  bool finished = gamma_wait == nullptr;                 // (a caller with an event of its own launches den_finish_kernel itself)
  hipError_t e = run_den_launches(a, resident_slot_rows, occupancy, st, why, gamma_wait, nullptr, &finished);
  if (e == hipSuccess && (a.phase_mask & 1) && !finished) e = launch_den_finish(a, st);   // objf (+ the check) from the stored totals
  return e;
}
}  // namespace

// ---- the burn-in controller state of a plan (DenArgs::tstate): a caller-owned device blob, attached by plan address
namespace {
std::mutex g_tstate_lock;
std::map<const void*, void*>& tstate_table() { static std::map<const void*, void*> t; return t; }
int32_t* tstate_of(const void* plans_dev) {
  std::lock_guard<std::mutex> guard(g_tstate_lock);
  auto it = tstate_table().find(plans_dev);
  return it == tstate_table().end() ? nullptr : (int32_t*)it->second;
}
}  // namespace
extern "C" int pychain_hip_den_tseg_state(const void* plans_dev, void* state_dev) {
  if (!plans_dev) return fail(PYCHAIN_HIP_EINVAL, "den_tseg_state: null plan");
  if (state_dev && ((uintptr_t)state_dev & 3)) return fail(PYCHAIN_HIP_EINVAL, "den_tseg_state: the state must be 4-byte aligned");
  std::lock_guard<std::mutex> guard(g_tstate_lock);
  if (state_dev) tstate_table()[plans_dev] = state_dev; else tstate_table().erase(plans_dev);
  return PYCHAIN_HIP_OK;
}
extern "C" size_t pychain_hip_den_tseg_state_bytes(void) { return kTsegStateWords * sizeof(int32_t); }

extern "C" int pychain_hip_den_uses_row_buffer(int64_t plan_stride_bytes, int resident_slot_rows, int H, int D, int B, int T,
                                              int input_is_exp) {
  if (B <= 0 || T <= 0 || H <= 0 || D <= 0 || resident_slot_rows == PYCHAIN_HIP_HINT_GENERAL) return 0;
  DenArgs a;
  memset(&a, 0, sizeof(a));
  a.plan_stride = plan_stride_bytes; a.B = B; a.T = T; a.D = D; a.H = H; a.Hp = roundup64(H); a.input_is_exp = input_is_exp ? 1 : 0;
  a.knobs = call_knobs();
  a.lazy = den_call_is_lazy(a, resident_slot_rows) ? 1 : 0;
  a.shape = a.lazy ? den_call_shape(a, resident_slot_rows) : 0;
  a.pair = den_call_is_pair(a, resident_slot_rows) ? 1 : 0;
  a.sg = (a.lazy && a.shape == kShapeDma && den_sg_eligible(a, resident_slot_rows)) ? 1 : 0;
  return den_would_exp_rows_ahead(a) ? 1 : 0;
}

extern "C" int pychain_hip_den_time_segments(int64_t plan_stride_bytes, int resident_slot_rows, int H, int D, int B, int T, int fused) {
  if (B <= 0 || T <= 0 || H <= 0 || D <= 0 || resident_slot_rows == PYCHAIN_HIP_HINT_GENERAL) return 1;
  DenArgs a;
  memset(&a, 0, sizeof(a));
  a.plan_stride = plan_stride_bytes; a.B = B; a.T = T; a.D = D; a.H = H; a.Hp = roundup64(H); a.frames_per_block = 32;
  a.knobs = call_knobs();
  a.fused = fused ? 1 : 0;
  a.check_all = a.knobs.verbose >= 1 ? 1 : 0;
  a.lazy = den_call_is_lazy(a, resident_slot_rows) ? 1 : 0;
  a.shape = a.lazy ? den_call_shape(a, resident_slot_rows) : 0;
  a.pair = den_call_is_pair(a, resident_slot_rows) ? 1 : 0;
  a.sg = (a.lazy && a.shape == kShapeDma && den_sg_eligible(a, resident_slot_rows)) ? 1 : 0;
  return den_time_segments(a, fused != 0);
}
namespace {
int den_half_native_q(int64_t plan_stride_bytes, int resident_slot_rows, int H, int D, int B, int T, int fused) {
  if (B <= 0 || T <= 0 || H <= 0 || D <= 0) return 0;
  DenArgs a;
  memset(&a, 0, sizeof(a));
  a.plan_stride = plan_stride_bytes; a.B = B; a.T = T; a.D = D; a.H = H; a.Hp = roundup64(H); a.frames_per_block = 32;
  a.knobs = call_knobs();
  a.fused = fused;
  return den_call_half_native(a, resident_slot_rows) ? 1 : 0;
}
}  // namespace
extern "C" int pychain_hip_den_half_native(int64_t plan_stride_bytes, int resident_slot_rows, int H, int D, int B, int T) {
  return den_half_native_q(plan_stride_bytes, resident_slot_rows, H, D, B, T, 0);
}
extern "C" int pychain_hip_num_half_native(int H, int K, int D) {
  if (H <= 0 || K <= 0 || D <= 0) return 0;
  NumArgs n;
  memset(&n, 0, sizeof(n));
  n.H = H; n.K = K; n.D = D; n.general = num_needs_general(H, K, D) ? 1 : 0; n.compat = call_knobs().num_compat;
  return num_half_native(n) ? 1 : 0;
}
extern "C" int pychain_hip_chain_loss_half_native(int64_t plan_stride_bytes, int resident_slot_rows, int den_H, int D, int B, int T,
                                                  int num_H, int num_K) {
  if (!den_half_native_q(plan_stride_bytes, resident_slot_rows, den_H, D, B, T, 1) || !pychain_hip_num_half_native(num_H, num_K, D)) return 0;
  DenArgs a;
  memset(&a, 0, sizeof(a));
  a.plan_stride = plan_stride_bytes; a.B = B; a.T = T; a.D = D; a.H = den_H; a.Hp = roundup64(den_H); a.frames_per_block = 32;
  a.knobs = call_knobs();
  a.fused = 1;
  a.fold_rows = (const float*)1;                          // (the fold's extra LDS counts: gamma2_lds_bytes)
  return den_uses_gamma2(a, (D + 63) / 64, resident_slot_rows) ? 1 : 0;
}

// ---- test hook: a long-lived kernel that pins CUs on another stream of the same process, as the channels of an overlapped
// RCCL bucket all-reduce do during DDP's backward: `workgroups` workgroups of 1024 threads and 100 KB of LDS
// (one to a CU), each sleeping for `microseconds`
namespace {
__global__ __launch_bounds__(1024) void occupy_kernel(unsigned long long ticks, int* sink) {
  extern __shared__ char occ_lds[];
  const unsigned long long t0 = wall_clock64();                   // 100 MHz
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (sink && threadIdx.x == 0 && occ_lds[threadIdx.x] == 77) *sink = 1;
}
}  // namespace
extern "C" int pychain_hip_debug_occupy(int workgroups, int microseconds, void* stream) {
  if (workgroups <= 0 || microseconds < 0 || microseconds > 2000000) return fail(PYCHAIN_HIP_EINVAL, "debug_occupy: bad arguments");
  const int lds = 100 * 1024;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
    return fail(PYCHAIN_HIP_ELAUNCH, "debug_occupy: cannot set the LDS size");
  hipLaunchKernelGGL(occupy_kernel, dim3(workgroups), dim3(1024), lds, (hipStream_t)stream, (unsigned long long)microseconds * 100ull, (int*)nullptr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PYCHAIN_HIP_ELAUNCH, "debug_occupy: %s", hipGetErrorString(e));
  return PYCHAIN_HIP_OK;
}

extern "C" int pychain_hip_debug_stream_rings(int T, int32_t* out, int out_len, int32_t* report_due, int due_len) {
  if (T <= 0 || out_len < 0 || due_len < 0 || (out_len && !out) || (due_len && !report_due))
    return fail(PYCHAIN_HIP_EINVAL, "debug_stream_rings: bad arguments");
  const int n = stream_ring_count(T);
  for (int r = 0; r < n && 2 * r + 1 < out_len; r++) {
    int lo, hi;
    stream_ring(T, r, lo, hi);
    out[2 * r] = lo; out[2 * r + 1] = hi;
  }
  for (int d = 0; d < due_len; d++) report_due[d] = stream_report_due(T, d) ? 1 : 0;
  return n;
}

extern "C" int pychain_hip_debug_launch_map(int T, int L, int t, int frames_per_block, int nseg,
                                            const int32_t* seg_bound, int seg, int32_t* out, int out_len) {
  if (T <= 0 || L <= 0 || L > T || frames_per_block <= 0 || nseg < 0 || nseg > 16 || (nseg && !seg_bound) || seg < 0 || (nseg && seg >= nseg))
    return fail(PYCHAIN_HIP_EINVAL, "debug_launch_map: bad arguments");
  return den_debug_launch_map(T, L, t, frames_per_block, nseg, seg_bound, seg, out, out_len);
}

extern "C" int pychain_hip_den_forward_backward(
    const void* plans_dev, int64_t plan_stride_bytes, int resident_slot_rows, int H, int D,
    const void* nnet_output, int nnet_output_dtype, int input_is_exp, const int64_t* seq_lengths,
    int B, int T, float leaky_hmm_coefficient, float grad_scale,
    float* objf_per_seq, void* grad, int32_t* bad_count, float* totals,
    void* workspace, size_t workspace_bytes, void* stream) {
  DenArgs a;
  int rc = fill_den_args(a, plans_dev, plan_stride_bytes, resident_slot_rows, H, D, nnet_output, nnet_output_dtype, input_is_exp, seq_lengths, B, T,
                         leaky_hmm_coefficient, grad_scale, objf_per_seq, grad, bad_count, workspace,
                         workspace_bytes, "den_forward_backward");
  if (rc != PYCHAIN_HIP_OK) return rc;
  a.loss_out = totals;                                 // (sum of the per-sequence objectives, frames, bad count: den_finish_kernel)
  if (a.x_half && !den_call_half_native(a, resident_slot_rows))
    return fail(PYCHAIN_HIP_EUNSUPPORTED, "den_forward_backward: this shape does not take 2-byte network outputs (pychain_hip_den_half_native)");
  hipStream_t st = (hipStream_t)stream;
  if (launch_zero_words(bad_count, 1, a.seq_progress, den_counter_words(a), st) != hipSuccess)
    return fail(PYCHAIN_HIP_ELAUNCH, "den_forward_backward: cannot zero the counters");
  const char* why = nullptr;
  hipError_t e = run_den(a, resident_slot_rows, true, st, &why);
  if (e != hipSuccess)
    return fail(why ? PYCHAIN_HIP_EUNSUPPORTED : PYCHAIN_HIP_ELAUNCH, "den_forward_backward: %s",
                why ? why : hipGetErrorString(e));
  return PYCHAIN_HIP_OK;
}

namespace {
struct NumCarve { size_t alpha, beta, logp, rows, upd, ucount, uidx, frac, gacc, compat, total; };
NumCarve num_carve(int B, int T, int H, int K, int D, bool compat) {
  NumCarve c;
  c.alpha = 0;
  c.beta = c.alpha + align256(8 * (size_t)B * (T + 1) * H);
  c.logp = c.beta + align256(8 * (size_t)B * (T + 1) * H);
  c.rows = c.logp + align256(8 * (size_t)B);
  c.upd = c.rows + align256(4 * (size_t)B * T * K);
  c.ucount = c.upd + align256(4 * (size_t)B * K);
  c.uidx = c.ucount + align256(4 * (size_t)B);
  c.frac = c.uidx + align256(4 * (size_t)B * K);
  c.gacc = c.frac + align256(4 * (size_t)B * T * K);           // general kernels only (num_general.hip): accumulator rows
  c.compat = c.gacc + (num_needs_general(H, K, D) ? align256(num_general_acc_bytes(D)) : 0);
  c.total = c.compat + (compat ? (size_t)B * num_compat_stride(H, K) : 0) + 256;   // (option num_compat: num_compat.hip)
  return c;
}

int fill_num_args(NumArgs& a, const int32_t* ft, const int32_t* fi, const float* fp,
                  const int32_t* bt, const int32_t* bi, const float* bp,
                  const float* initial, const float* final_, int graph_batch_stride,
                  const void* nnet_output, int x_dtype, const int64_t* seq_lengths,
                  int B, int T, int D, int H, int K, int grad_mode, float grad_scale,
                  float* objf_per_seq, void* grad, int32_t* bad_count,
                  void* workspace, size_t workspace_bytes, const char* who) {
  if (x_dtype < PYCHAIN_HIP_F32 || x_dtype > PYCHAIN_HIP_F16)
    return fail(PYCHAIN_HIP_EINVAL, "%s: unknown nnet_output_dtype %d", who, x_dtype);
  if (!ft || !fi || !fp || !bt || !bi || !bp || !initial || !final_ || !nnet_output || !seq_lengths ||
      !objf_per_seq || !grad || !bad_count || !workspace)
    return fail(PYCHAIN_HIP_EINVAL, "%s: null pointer argument", who);
  if (B <= 0 || T <= 0 || H <= 0 || D <= 0 || K <= 0)
    return fail(PYCHAIN_HIP_EINVAL, "%s: bad sizes B=%d T=%d H=%d K=%d D=%d", who, B, T, H, K, D);
  if (graph_batch_stride != 0 && graph_batch_stride != 1)
    return fail(PYCHAIN_HIP_EINVAL, "%s: graph_batch_stride must be 0 or 1", who);
  if (grad_mode < PYCHAIN_HIP_GRAD_LOG || grad_mode > PYCHAIN_HIP_GRAD_ACCUM)
    return fail(PYCHAIN_HIP_EINVAL, "%s: unknown grad_mode %d", who, grad_mode);
  if (((uintptr_t)nnet_output | (uintptr_t)grad | (uintptr_t)fi | (uintptr_t)bi) & 15)
    return fail(PYCHAIN_HIP_EINVAL, "%s: nnet_output, grad and index tensors must be 16-byte aligned", who);
  const CallKnobs knobs = call_knobs();
  const NumCarve c = num_carve(B, T, H, K, D, knobs.num_compat != 0);
  if (workspace_bytes < c.total) return fail(PYCHAIN_HIP_EWORKSPACE, "%s: workspace too small", who);
  memset(&a, 0, sizeof(a));
  a.fwd_trans = ft; a.fwd_idx = fi; a.fwd_probs = fp; a.bwd_trans = bt; a.bwd_idx = bi; a.bwd_probs = bp;
  a.initial = initial; a.final_ = final_; a.x = (const float*)nnet_output; a.x_half = x_dtype; a.lengths = seq_lengths;
  a.objf = objf_per_seq; a.grad = (float*)grad; a.bad = bad_count;
  a.graph_stride = graph_batch_stride; a.B = B; a.T = T; a.D = D; a.H = H; a.K = K;
  a.grad_mode = grad_mode; a.grad_scale = grad_scale; a.frames_per_block = 32;
  a.check_all = knobs.verbose >= 1 ? 1 : 0;
  if (knobs.corrupt_what == 2 && knobs.corrupt_b < B && knobs.corrupt_t < T) {
    a.corrupt_b = knobs.corrupt_b; a.corrupt_t = knobs.corrupt_t; a.corrupt_log = logf(knobs.corrupt_scale);
  } else a.corrupt_b = -1;
  a.watch_nan = 1;                                   // (the fused loss clears it: NumArgs::watch_nan)
  char* ws = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  a.alpha_ws = (double*)(ws + c.alpha); a.beta_ws = (double*)(ws + c.beta); a.logp_ws = (double*)(ws + c.logp);
  a.rows_ws = (float*)(ws + c.rows); a.upd_ws = (int32_t*)(ws + c.upd); a.ucount_ws = (int32_t*)(ws + c.ucount);
  a.uidx_ws = (int32_t*)(ws + c.uidx); a.frac_ws = (float*)(ws + c.frac);
  a.general = num_needs_general(H, K, D) ? 1 : 0;             // graphs beyond the tile kernels: num_general.hip
  a.gen_acc = ws + c.gacc;
  a.compat = knobs.num_compat; a.compat_ws = ws + c.compat; a.compat_stride = num_compat_stride(H, K);
  return PYCHAIN_HIP_OK;
}
}  // namespace

extern "C" size_t pychain_hip_num_workspace_bytes(int B, int T, int H, int K, int D) {
  if (B <= 0 || T <= 0 || H <= 0 || K <= 0 || D <= 0) return 0;
  return num_carve(B, T, H, K, D, call_knobs().num_compat != 0).total;   // (the calling thread's options, as the call will read them)
}

extern "C" int pychain_hip_num_forward_backward(
    const int32_t* ft, const int32_t* fi, const float* fp,
    const int32_t* bt, const int32_t* bi, const float* bp,
    const float* initial, const float* final_, int graph_batch_stride,
    const void* nnet_output, int nnet_output_dtype, const int64_t* seq_lengths,
    int B, int T, int D, int H, int K, int grad_mode, float grad_scale,
    float* objf_per_seq, float* grad, int32_t* bad_count,
    void* workspace, size_t workspace_bytes, void* stream) {
  NumArgs a;
  int rc = fill_num_args(a, ft, fi, fp, bt, bi, bp, initial, final_, graph_batch_stride, nnet_output, nnet_output_dtype, seq_lengths,
                         B, T, D, H, K, grad_mode, grad_scale, objf_per_seq, grad, bad_count, workspace,
                         workspace_bytes, "num_forward_backward");
  if (rc != PYCHAIN_HIP_OK) return rc;
  // (2-byte network outputs are READ as they are by the tile recursions; the gradient of this entry point stays fp32)
  if (a.x_half && !num_half_native(a))
    return fail(PYCHAIN_HIP_EUNSUPPORTED, "num_forward_backward: this shape does not take 2-byte network outputs (pychain_hip_num_half_native)");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(bad_count, 0, sizeof(int32_t), st) != hipSuccess)
    return fail(PYCHAIN_HIP_ELAUNCH, "num_forward_backward: hipMemsetAsync failed");
  const char* why = nullptr;
  hipError_t e = hipSuccess;
  if (a.compat) {                                        // the reference's own arithmetic, one launch (num_compat.hip)
    e = launch_num_compat(a, 3, st);
  } else {
    e = launch_num_fb(a, st, &why);
    if (e == hipSuccess && a.corrupt_b >= 0) e = launch_num_corrupt(a, st);
    if (e == hipSuccess) e = launch_num_occ(a, false, st, &why);
  }
  if (e != hipSuccess)
    return fail(why ? PYCHAIN_HIP_EUNSUPPORTED : PYCHAIN_HIP_ELAUNCH, "num_forward_backward: %s",
                why ? why : hipGetErrorString(e));
  return PYCHAIN_HIP_OK;
}

// ---- fused ChainLoss ------------------------------------------------------------------

namespace {
// one fused call over the B sequences it is given (pychain_hip_chain_loss_forward: all of them, or one slice of a large batch)
int chain_loss_forward_one(
    const void* plans_dev, int64_t plan_stride_bytes, int resident_slot_rows, int den_H, float leaky,
    const int32_t* ft, const int32_t* fi, const float* fp, const int32_t* bt, const int32_t* bi, const float* bp,
    const float* initial, const float* final_, int graph_batch_stride, int num_H, int num_K,
    const void* nnet_output, int nnet_output_dtype, const int64_t* seq_lengths, int B, int T, int D,
    float* den_objf, float* num_objf, void* grad, float grad_scale, int32_t* bad_count,
    float loss_scale, const float* loss_norm_dev, float* totals,
    void* den_ws, size_t den_ws_bytes, void* num_ws, size_t num_ws_bytes, void* stream) {
  const char* who = "chain_loss_forward";
  if (!bad_count) return fail(PYCHAIN_HIP_EINVAL, "%s: null bad_count", who);
  DenArgs da;
  // without `grad` only the recursions run; any non-null aligned pointer then passes the checks
  int rc = fill_den_args(da, plans_dev, plan_stride_bytes, resident_slot_rows, den_H, D, nnet_output, nnet_output_dtype, 0, seq_lengths, B, T, leaky,
                         grad_scale, den_objf, grad ? grad : den_ws, bad_count, den_ws, den_ws_bytes, who);
  if (rc != PYCHAIN_HIP_OK) return rc;
  NumArgs na;
  rc = fill_num_args(na, ft, fi, fp, bt, bi, bp, initial, final_, graph_batch_stride, nnet_output, nnet_output_dtype, seq_lengths,
                     B, T, D, num_H, num_K, PYCHAIN_HIP_GRAD_ACCUM, -grad_scale, num_objf,
                     grad ? grad : num_ws, bad_count + 1, num_ws, num_ws_bytes, who);
  if (rc != PYCHAIN_HIP_OK) return rc;

  da.fused = 1;
  na.watch_nan = 0;              // the denominator's alpha workgroups watch every element of every row (NumArgs::watch_nan)
  // the scalars of ChainLoss.forward from den_finish_kernel's last workgroup (DenArgs::loss_out)
  da.loss_out = totals; da.loss_num_objf = num_objf; da.loss_scale = loss_scale; da.loss_norm_dev = loss_norm_dev; da.bad_words = 2;
  hipStream_t st = (hipStream_t)stream;
  SideStream* side = side_streams_for(st);
  if (!side) return fail(PYCHAIN_HIP_ELAUNCH, "%s: cannot create the side stream", who);
  const char* why = nullptr;
  // a device-side normaliser of the loss (loss_norm_dev: ChainLoss(avg=True) with the lengths on the device) also divides the
  // gradient written here: its reciprocal goes into a spare counter word at the head of the call and the occupancy launches
  // of both sides read it (DenArgs / NumArgs::grad_scale_dev) - not a pass over [B, T, D] behind the call
  float* inv_norm = (loss_norm_dev && grad) ? reinterpret_cast<float*>(da.progress + 56) : nullptr;
  if (inv_norm) { da.grad_scale_dev = inv_norm; na.grad_scale_dev = inv_norm; }
  hipError_t e = launch_zero_words(bad_count, 2, da.seq_progress, den_counter_words(da), st, loss_norm_dev, inv_norm);
  // The two-frame occupancy kernel folds the numerator in (grad = scale * (gamma_den - gamma_num), written
  // once): the numerator then also produces compact occupancy rows on its stream, and the occupancy
  // launches wait for them.  Otherwise the numerator is accumulated into the gradient afterwards.
  // (decided with the fold's own LDS rows counted in: gamma2_lds_bytes)
  da.fold_rows = na.rows_ws;
  const bool fold = grad && resident_slot_rows != PYCHAIN_HIP_HINT_GENERAL && !na.general && !na.compat &&
                    den_uses_gamma2(da, (D + 63) / 64, resident_slot_rows);
  if (fold) {
    da.fold_upd = na.upd_ws; da.fold_ucount = na.ucount_ws; da.fold_K = num_K;
    da.fold_scale = -grad_scale;
  } else da.fold_rows = nullptr;
  // 2-byte network outputs: both sides take them and the gradient is written once, by the fold (chain_loss_half_native)
  if (nnet_output_dtype != PYCHAIN_HIP_F32 && !(den_call_half_native(da, resident_slot_rows) && num_half_native(na) && (fold || !grad)))
    return fail(PYCHAIN_HIP_EUNSUPPORTED, "%s: this shape does not take 2-byte network outputs (pychain_hip_chain_loss_half_native)", who);
  // fork: numerator on the side stream, denominator recursion on the caller's stream
  if (e == hipSuccess) e = hipEventRecord(side->fork, st);
  if (e == hipSuccess) e = hipStreamWaitEvent(side->stream, side->fork, 0);
  // (option num_compat: the numerator is ONE launch behind the denominator's occupancy launches - it accumulates into the
  // gradient they wrote - and in front of den_finish_kernel, which reads its log-probabilities)
  if (e == hipSuccess && grad && !na.compat) e = launch_num_prep(na, side->stream, &why);
  if (e == hipSuccess && !na.compat) e = launch_num_fb(na, side->stream, &why);
  if (e == hipSuccess && na.corrupt_b >= 0 && !na.compat) e = launch_num_corrupt(na, side->stream);
  if (e == hipSuccess && grad && !na.general && !na.compat) e = launch_num_occ(na, true, side->stream, &why);   // compact rows, off the critical path
  if (e == hipSuccess) e = hipEventRecord(side->join, side->stream);
  da.phase_mask = da.knobs.den_phase_mask == 0 ? 0 : 3;     // (mask 0: only the numerator's launches - a measurement aid, outputs not meaningful)
  // (den_finish_kernel reads the numerator's objectives and its bad count for `totals`: the join precedes it)
  // (den_finish_kernel needs the numerator's objectives and bad count: early only where the occupancy launch waits for them)
  bool finished = fold;
  if (e == hipSuccess) e = run_den_launches(da, resident_slot_rows, grad != nullptr, st, &why, fold ? side->join : nullptr, side->fork, &finished);
  // join (a call whose occupancy launch folded the numerator in has it behind it already: that launch waited for the numerator's
  // event and the caller's stream for that launch - one barrier packet less, ~5 us, between it and den_finish_kernel)
  const bool joined = fold && da.phase_mask == 3;     // (every schedule of run_den_launches puts the wait in front of its first occupancy launch)
  if (e == hipSuccess && !joined) e = hipStreamWaitEvent(st, side->join, 0);
  if (e == hipSuccess && na.compat) e = launch_num_compat(na, grad ? 3 : 1, st);            // grad -= grad_scale * gamma_num, the reference's way
  if (e == hipSuccess && (da.phase_mask & 1) && !finished) e = launch_den_finish(da, st);     // (phase mask 0: the numerator alone - bench.py times it so)
  if (e == hipSuccess && grad && !fold && !na.compat)                               // grad -= grad_scale * gamma_num
    e = na.general ? launch_num_occ(na, false, st, &why) : launch_num_scatter(na, st, &why);
  if (e != hipSuccess)
    return fail(why ? PYCHAIN_HIP_EUNSUPPORTED : PYCHAIN_HIP_ELAUNCH, "%s: %s", who, why ? why : hipGetErrorString(e));
  return PYCHAIN_HIP_OK;
}

// ---- a batch larger than the chip: slices (round 5).  From B = 224 on 256 CUs the recursion workgroups of ONE call (two
// sequences each: den_recursion_pair_kernel) hold every CU for the whole recursion, the numerator only finds room as they
// end, and the occupancy launch that folds it in starts late (DESIGN.md §4 "B >= 128": B = 256 11.1 ms against 2 x 5.06 for
// two calls of 128).  The fused call that also writes the gradient is therefore run over slices of about half the CU count
// in sequences, one after the other on the caller's stream, in the SAME workspaces; every slice reports into a scratch
// line at the end of the denominator's workspace, and one small launch sums the bad counts and forms `totals` from the
// per-sequence log-probabilities of the whole batch exactly as den_finish_kernel does (fp64, rounded once).
// Option chain_slices: 0 / 1 never, n >= 2 that many (where the batch allows), default automatic.
struct SliceLine { int32_t bad[2]; int32_t pad[2]; float totals[PYCHAIN_HIP_TOTALS]; };     // 48 bytes per slice
constexpr int kMaxSlices = 32;
__global__ void chain_slices_combine_kernel(const SliceLine* lines, int nslices, const float* den_objf, const float* num_objf,
                                            const int64_t* lengths, int B, int T, float loss_scale, const float* loss_norm_dev,
                                            int32_t* bad_count, float* totals) {
  __shared__ double part[2][4];
  double acc = 0.0, frames = 0.0;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    acc += (double)den_objf[i] - (double)num_objf[i];
    const int64_t l = lengths[i];
    frames += (double)(l < 1 ? 1 : (l > T ? T : l));
  }
  for (int o = 32; o > 0; o >>= 1) { acc += __shfl_down(acc, o); frames += __shfl_down(frames, o); }
  if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = acc; part[1][threadIdx.x >> 6] = frames; }
  __syncthreads();
  if (threadIdx.x == 0) {
    acc = (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]);
    frames = (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]);
    int b0 = 0, b1 = 0;
    float redone = 0.f, segs = 1.f, worst = 0.f;
    for (int c = 0; c < nslices; c++) {
      b0 += lines[c].bad[0]; b1 += lines[c].bad[1];
      redone += lines[c].totals[5]; segs = fmaxf(segs, lines[c].totals[6]); worst = fmaxf(worst, lines[c].totals[7]);
    }
    bad_count[0] = b0; bad_count[1] = b1;
    if (totals) {
      double t = acc * (double)loss_scale;
      if (loss_norm_dev) t /= (double)*loss_norm_dev;
      totals[0] = (float)t; totals[1] = (float)frames; totals[2] = (float)(b0 + b1); totals[3] = (float)acc; totals[4] = (float)t;
      totals[5] = redone; totals[6] = segs; totals[7] = worst;
    }
  }
}
// slices of a fused call over B sequences (1 = the call as it is)
int chain_loss_slices(int B, int resident_slot_rows, int64_t plan_stride_bytes, bool with_grad) {
  const int want = call_knobs().chain_slices;
  if (!with_grad || want == 0 || want == 1 || resident_slot_rows == PYCHAIN_HIP_HINT_GENERAL || plan_stride_bytes != 0) return 1;
  const int cus = device_cu_count();
  int n = 1;
  if (want >= 2) n = want;
  // automatic: slices of about CUs / 2 sequences, rather a few more than one slice more (measured on 256 CUs, profiles/r05_slices.txt:
  // B = 320 as 2 x 160 13.2 ms, as 3 x 107 13.8, one call 13.7; B = 384 as 3 x 128 15.6, as 2 x 192 17.0, one call 16.1)
  else if (8 * B >= 7 * cus) n = (int)(2.0 * B / cus + 0.25);
  n = std::min(n, std::min(kMaxSlices, B / 2));
  return std::max(n, 1);
}
}  // namespace

namespace {
// Which denominator workspaces were last written by a SLICED fused forward (every slice overwrote the one before it: only
// the last slice's trajectories are there, so pychain_hip_chain_loss_backward on them would write a wrong gradient for all
// earlier slices - ADVICE r5).  Host-side bookkeeping by workspace address, updated by every pychain_hip_chain_loss_forward.
std::mutex g_sliced_lock;
std::map<const void*, bool>& sliced_workspaces() { static std::map<const void*, bool> m; return m; }
void note_forward_workspace(const void* den_ws, bool sliced) {
  std::lock_guard<std::mutex> guard(g_sliced_lock);
  auto& m = sliced_workspaces();
  if (sliced) { if (m.size() > 256) m.clear(); m[den_ws] = true; } else m.erase(den_ws);
}
bool workspace_was_sliced(const void* den_ws) {
  std::lock_guard<std::mutex> guard(g_sliced_lock);
  return sliced_workspaces().count(den_ws) != 0;
}
}  // namespace

extern "C" int pychain_hip_chain_loss_slices(int64_t plan_stride_bytes, int resident_slot_rows, int B) {
  return chain_loss_slices(B, resident_slot_rows, plan_stride_bytes, true);
}

extern "C" int pychain_hip_chain_loss_forward(
    const void* plans_dev, int64_t plan_stride_bytes, int resident_slot_rows, int den_H, float leaky,
    const int32_t* ft, const int32_t* fi, const float* fp, const int32_t* bt, const int32_t* bi, const float* bp,
    const float* initial, const float* final_, int graph_batch_stride, int num_H, int num_K,
    const void* nnet_output, int nnet_output_dtype, const int64_t* seq_lengths, int B, int T, int D,
    float* den_objf, float* num_objf, void* grad, float grad_scale, int32_t* bad_count,
    float loss_scale, const float* loss_norm_dev, float* totals,
    void* den_ws, size_t den_ws_bytes, void* num_ws, size_t num_ws_bytes, void* stream) {
  const int nsl = (B > 0 && T > 0 && D > 0 && den_ws && bad_count && den_objf && num_objf && seq_lengths && nnet_output && ft && fi && fp &&
                   bt && bi && bp && initial && final_ && den_ws_bytes > 8192)
                      ? chain_loss_slices(B, resident_slot_rows, plan_stride_bytes, grad != nullptr) : 1;
  note_forward_workspace(den_ws, nsl > 1);
  if (nsl <= 1)
    return chain_loss_forward_one(plans_dev, plan_stride_bytes, resident_slot_rows, den_H, leaky, ft, fi, fp, bt, bi, bp, initial, final_,
                                  graph_batch_stride, num_H, num_K, nnet_output, nnet_output_dtype, seq_lengths, B, T, D, den_objf, num_objf,
                                  grad, grad_scale, bad_count, loss_scale, loss_norm_dev, totals, den_ws, den_ws_bytes, num_ws, num_ws_bytes,
                                  stream);
  // the scratch lines: the last 4 KiB of the denominator's workspace (sized by the caller for all B sequences; a slice
  // needs about 1 / nsl of it)
  const size_t ws_bytes = (den_ws_bytes - 4096) & ~(size_t)255;
  SliceLine* lines = reinterpret_cast<SliceLine*>((char*)den_ws + ws_bytes);
  const size_t esz = nnet_output_dtype == PYCHAIN_HIP_F32 ? 4 : 2;
  // (a multiple of 8 sequences: every slice's rows, gradient and graph tensors start 16-byte aligned whatever T and D are -
  // 2-byte rows of odd T x D included -, and the pair recursion takes its sequences two by two)
  const int per = ((B + nsl - 1) / nsl + 7) & ~7;
  int c = 0;
  for (int b0 = 0; b0 < B; b0 += per, c++) {
    const int nb = std::min(per, B - b0);
    const size_t g = graph_batch_stride ? (size_t)b0 : 0;     // per-sequence numerator graphs: [G, K, 3] / [G, H, 2] / [G, K] / [G, H]
    const int rc = chain_loss_forward_one(
        plans_dev, plan_stride_bytes, resident_slot_rows, den_H, leaky,
        ft + g * num_K * 3, fi + g * num_H * 2, fp + g * num_K, bt + g * num_K * 3, bi + g * num_H * 2, bp + g * num_K,
        initial + g * num_H, final_ + g * num_H, graph_batch_stride, num_H, num_K,
        (const char*)nnet_output + (size_t)b0 * T * D * esz, nnet_output_dtype, seq_lengths + b0, nb, T, D,
        den_objf + b0, num_objf + b0, (char*)grad + (size_t)b0 * T * D * esz, grad_scale, lines[c].bad,
        loss_scale, loss_norm_dev, lines[c].totals, den_ws, ws_bytes, num_ws, num_ws_bytes, stream);
    if (rc != PYCHAIN_HIP_OK) return rc;
  }
  hipLaunchKernelGGL(chain_slices_combine_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, lines, c, den_objf, num_objf, seq_lengths,
                     B, T, loss_scale, loss_norm_dev, bad_count, totals);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PYCHAIN_HIP_ELAUNCH, "chain_loss_forward: %s", hipGetErrorString(e));
  return PYCHAIN_HIP_OK;
}

// grad *= *scale_dev unless the scalar is exactly 1 (the common `loss.backward()` case costs one
// tiny launch; any other upstream gradient costs one pass over the buffer).
namespace {
__global__ void rescale_kernel(float4* data, size_t n4, float* tail, int ntail, const float* scale_dev) {
  const float g = *scale_dev;
  if (g == 1.0f) return;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = data[i];
    v.x *= g; v.y *= g; v.z *= g; v.w *= g;
    data[i] = v;
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] *= g;
}
__global__ void rescale_half_kernel(uint32_t* data, size_t n2, uint16_t* tail, const float* scale_dev, int bf16) {
  const float g = *scale_dev;
  if (g == 1.0f) return;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    float lo, hi;
    half2_to_f32(data[i], bf16 != 0, lo, hi);
    data[i] = pack_half2(lo * g, hi * g, bf16 != 0);
  }
  if (tail && blockIdx.x == 0 && threadIdx.x == 0) *tail = (uint16_t)f32_to_half_bits(half_bits_to_f32(*tail, bf16 != 0) * g, bf16 != 0);
}
}  // namespace

extern "C" int pychain_hip_rescale(void* data_, int dtype, size_t n, const float* scale_dev, void* stream) {
  if (!data_ || !scale_dev || ((uintptr_t)data_ & 15) || dtype < PYCHAIN_HIP_F32 || dtype > PYCHAIN_HIP_F16)
    return fail(PYCHAIN_HIP_EINVAL, "rescale: null or unaligned argument, or unknown dtype");
  if (dtype != PYCHAIN_HIP_F32) {                        // 2-byte gradients: value * scale in fp32, rounded to nearest even
    hipLaunchKernelGGL(rescale_half_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (uint32_t*)data_, n / 2,
                       (n & 1) ? (uint16_t*)data_ + (n - 1) : nullptr, scale_dev, dtype == PYCHAIN_HIP_BF16 ? 1 : 0);
    hipError_t eh = hipGetLastError();
    if (eh != hipSuccess) return fail(PYCHAIN_HIP_ELAUNCH, "rescale: %s", hipGetErrorString(eh));
    return PYCHAIN_HIP_OK;
  }
  float* data = (float*)data_;
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(rescale_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (float4*)data, n4,
                     data + 4 * n4, (int)(n - 4 * n4), scale_dev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PYCHAIN_HIP_ELAUNCH, "rescale: %s", hipGetErrorString(e));
  return PYCHAIN_HIP_OK;
}

// out[0] = (sum den_objf - sum num_objf) * scale [/ *norm_dev]: the scalar of ChainLoss.forward (loss.py:100-104) in ONE
// launch instead of two reductions, a subtraction and a scaling of the host framework (each ~6 us, launch-bound, in the
// serial tail of every step).  fp64 accumulation, one rounding.
namespace {
__global__ void loss_total_kernel(const float* den, const float* num, int B, float scale, const float* norm_dev, float* out) {
  __shared__ double part[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < B; i += blockDim.x) acc += (double)den[i] - (num ? (double)num[i] : 0.0);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = ((part[0] + part[1]) + (part[2] + part[3])) * (double)scale;
    if (norm_dev) t /= (double)*norm_dev;
    out[0] = (float)t;
  }
}
}  // namespace

extern "C" int pychain_hip_loss_total(const float* den_objf_per_seq, const float* num_objf_per_seq, int B, float scale,
                                      const float* norm_dev, float* out, void* stream) {
  if (!den_objf_per_seq || !out || B <= 0) return fail(PYCHAIN_HIP_EINVAL, "loss_total: null pointer or empty batch");
  hipLaunchKernelGGL(loss_total_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, den_objf_per_seq, num_objf_per_seq, B, scale,
                     norm_dev, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PYCHAIN_HIP_ELAUNCH, "loss_total: %s", hipGetErrorString(e));
  return PYCHAIN_HIP_OK;
}

namespace {
int chain_loss_backward_impl(
    const void* plans_dev, int64_t plan_stride_bytes, int resident_slot_rows, int den_H,
    const int32_t* ft, const int32_t* fi, const float* fp, int graph_batch_stride, int num_H, int num_K,
    const void* nnet_output, int nnet_output_dtype, const int64_t* seq_lengths, int B, int T, int D,
    float grad_scale, const float* grad_scale_dev, void* grad, int32_t* bad_count,
    void* den_ws, size_t den_ws_bytes, void* num_ws, size_t num_ws_bytes, void* stream, bool zero_bad) {
  const char* who = "chain_loss_backward";
  if (!bad_count || !ft || !fi || !fp) return fail(PYCHAIN_HIP_EINVAL, "%s: null pointer argument", who);
  if (workspace_was_sliced(den_ws))
    return fail(PYCHAIN_HIP_EUNSUPPORTED, "%s: the forward call on these workspaces wrote its gradient in slices (a batch larger than the "
                "chip: pychain_hip_chain_loss_slices) - the workspaces hold the last slice's trajectories only.  Take the gradient "
                "that call wrote, or call pychain_hip_chain_loss_forward without a gradient first", who);
  DenArgs da;
  float dummy_coef = 0.5f;       // the occupancy launch does not use the leaky coefficient
  if (nnet_output_dtype != PYCHAIN_HIP_F32)              // (the numerator's occupancy launch accumulates into an fp32 gradient)
    return fail(PYCHAIN_HIP_EUNSUPPORTED, "%s: 2-byte network outputs are taken by the forward call that also writes the gradient", who);
  int rc = fill_den_args(da, plans_dev, plan_stride_bytes, resident_slot_rows, den_H, D, nnet_output, nnet_output_dtype, 0, seq_lengths, B, T, dummy_coef,
                         grad_scale, (float*)den_ws, grad, bad_count, den_ws, den_ws_bytes, who);
  if (rc != PYCHAIN_HIP_OK) return rc;
  da.grad_scale_dev = grad_scale_dev;
  da.fused = 1;                                          // (the forward call that stored the rows decided as a fused call)
  da.lazy = den_call_is_lazy(da, resident_slot_rows) ? 1 : 0;
  da.shape = da.lazy ? den_call_shape(da, resident_slot_rows) : 0;
  da.sg = (da.lazy && da.shape == kShapeDma && den_sg_eligible(da, resident_slot_rows)) ? 1 : 0;   // (what the stored alpha rows ARE: DenArgs::sg)
  NumArgs na;
  // the occupancy launch reads only the forward transitions / indices / log-probs of the graphs
  rc = fill_num_args(na, ft, fi, fp, ft, fi, fp, fp, fp,
                     graph_batch_stride, nnet_output, nnet_output_dtype, seq_lengths, B, T, D, num_H, num_K, PYCHAIN_HIP_GRAD_ACCUM,
                     -grad_scale, (float*)num_ws, grad, bad_count + 1, num_ws, num_ws_bytes, who);
  if (rc != PYCHAIN_HIP_OK) return rc;
  na.grad_scale_dev = grad_scale_dev;
  hipStream_t st = (hipStream_t)stream;
  const char* why = nullptr;
  hipError_t e = zero_bad ? hipMemsetAsync(bad_count, 0, 2 * sizeof(int32_t), st) : hipSuccess;
  da.phase_mask = 2;
  if (e == hipSuccess)
    e = resident_slot_rows == PYCHAIN_HIP_HINT_GENERAL ? launch_den_general(da, st) : launch_den(da, (D + 63) / 64, resident_slot_rows, st, &why);
  if (e == hipSuccess) e = na.compat ? launch_num_compat(na, 2, st) : launch_num_occ(na, false, st, &why);
  if (e != hipSuccess)
    return fail(why ? PYCHAIN_HIP_EUNSUPPORTED : PYCHAIN_HIP_ELAUNCH, "%s: %s", who, why ? why : hipGetErrorString(e));
  return PYCHAIN_HIP_OK;
}

}  // namespace

extern "C" int pychain_hip_chain_loss_backward(
    const void* plans_dev, int64_t plan_stride_bytes, int resident_slot_rows, int den_H,
    const int32_t* ft, const int32_t* fi, const float* fp, int graph_batch_stride, int num_H, int num_K,
    const void* nnet_output, int nnet_output_dtype, const int64_t* seq_lengths, int B, int T, int D,
    float grad_scale, const float* grad_scale_dev, void* grad, int32_t* bad_count,
    void* den_ws, size_t den_ws_bytes, void* num_ws, size_t num_ws_bytes, void* stream) {
  return chain_loss_backward_impl(plans_dev, plan_stride_bytes, resident_slot_rows, den_H, ft, fi, fp, graph_batch_stride,
                                  num_H, num_K, nnet_output, nnet_output_dtype, seq_lengths, B, T, D, grad_scale, grad_scale_dev, grad,
                                  bad_count, den_ws, den_ws_bytes, num_ws, num_ws_bytes, stream, true);
}

extern "C" int pychain_hip_chain_loss_forward_backward(
    const void* plans_dev, int64_t plan_stride_bytes, int resident_slot_rows, int den_H, float leaky,
    const int32_t* ft, const int32_t* fi, const float* fp, const int32_t* bt, const int32_t* bi, const float* bp,
    const float* initial, const float* final_, int graph_batch_stride, int num_H, int num_K,
    const void* nnet_output, int nnet_output_dtype, const int64_t* seq_lengths, int B, int T, int D, float grad_scale,
    float* den_objf, float* num_objf, void* grad, int32_t* bad_count,
    float loss_scale, const float* loss_norm_dev, float* totals,
    void* den_ws, size_t den_ws_bytes, void* num_ws, size_t num_ws_bytes, void* stream) {
  if (!grad) return fail(PYCHAIN_HIP_EINVAL, "chain_loss_forward_backward: null grad");
  return pychain_hip_chain_loss_forward(plans_dev, plan_stride_bytes, resident_slot_rows, den_H, leaky, ft, fi, fp, bt,
                                        bi, bp, initial, final_, graph_batch_stride, num_H, num_K, nnet_output, nnet_output_dtype,
                                        seq_lengths, B, T, D, den_objf, num_objf, grad, grad_scale, bad_count, loss_scale,
                                        loss_norm_dev, totals, den_ws, den_ws_bytes, num_ws, num_ws_bytes, stream);
}

// ---- on-device reorder of a staged batch (include/pychain_hip.h: batch containers) ------------------------------------
namespace {
struct GatherArgs { int64_t in_off[PYCHAIN_HIP_BATCH_FIELDS], out_off[PYCHAIN_HIP_BATCH_FIELDS], row_bytes[PYCHAIN_HIP_BATCH_FIELDS]; };
__global__ void batch_gather_kernel(const char* in, char* out, const int64_t* order, int B_in, GatherArgs g) {
  const int b = blockIdx.x, f = blockIdx.y;
  const int64_t rb = g.row_bytes[f];
  if (rb == 0) return;
  int64_t src = order[b];
  if (src < 0 || src >= B_in) src = 0;                                    // (validated on the host where the order lives there)
  const uint32_t* s = reinterpret_cast<const uint32_t*>(in + g.in_off[f] + rb * src);
  uint32_t* d = reinterpret_cast<uint32_t*>(out + g.out_off[f] + rb * b);
  for (int64_t i = threadIdx.x; i < rb / 4; i += blockDim.x) d[i] = s[i];
}
}  // namespace
extern "C" int pychain_hip_batch_reorder_dev(int B_in, int B_out, int K, int H, int log_domain, const void* in_dev, void* out_dev,
                                             const int64_t* order_dev, void* stream) {
  GatherArgs g;
  if (pychain_hip_batch_layout(B_in, K, H, log_domain, g.in_off, g.row_bytes) < 0 ||
      pychain_hip_batch_layout(B_out, K, H, log_domain, g.out_off, g.row_bytes) < 0)
    return PYCHAIN_HIP_EINVAL;
  if (!in_dev || !out_dev || !order_dev) return fail(PYCHAIN_HIP_EINVAL, "batch_reorder_dev: null pointer");
  hipLaunchKernelGGL(batch_gather_kernel, dim3(B_out, PYCHAIN_HIP_BATCH_FIELDS), dim3(256), 0, (hipStream_t)stream,
                     (const char*)in_dev, (char*)out_dev, order_dev, B_in, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PYCHAIN_HIP_ELAUNCH, "batch_reorder_dev: %s", hipGetErrorString(e));
  return PYCHAIN_HIP_OK;
}

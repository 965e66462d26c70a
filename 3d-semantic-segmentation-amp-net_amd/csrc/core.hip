// core.hip -- error plumbing and version entry points of the C ABI (include/ampnet_hip.h).
#include "common.h"
#include <algorithm>
#include <cstring>
#include <iterator>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace ampnet {

char *err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// ---- event profiler --------------------------------------------------------------------------------
namespace {
struct ProfEntry {
    char name[64];
    double flops, bytes;
    hipEvent_t e0, e1;
};
bool g_prof_on = false;
std::vector<ProfEntry> g_prof;
}  // namespace

ProfScope::ProfScope(const char *name, double flops, double bytes, hipStream_t s) : slot(-1), st(s)
{
    if (!g_prof_on) return;
    ProfEntry e;
    snprintf(e.name, sizeof(e.name), "%s", name);
    e.flops = flops;
    e.bytes = bytes;
    if (hipEventCreate(&e.e0) != hipSuccess || hipEventCreate(&e.e1) != hipSuccess) return;
    (void)hipEventRecord(e.e0, st);
    g_prof.push_back(e);
    slot = (int)g_prof.size() - 1;
}

ProfScope::~ProfScope()
{
    if (slot >= 0) (void)hipEventRecord(g_prof[slot].e1, st);
}

}  // namespace ampnet

extern "C" int ampnet_profile_enable(int on)
{
    for (auto &e : ampnet::g_prof) {
        (void)hipEventDestroy(e.e0);
        (void)hipEventDestroy(e.e1);
    }
    ampnet::g_prof.clear();
    ampnet::g_prof_on = on != 0;
    return AMPNET_OK;
}

// Aggregates the recorded launches by kernel name.  names: [max][64] chars; returns the number of rows written.
extern "C" int ampnet_profile_read(int max_rows, char *names, double *ms, long long *calls, double *flops, double *bytes)
{
    using namespace ampnet;
    if (hipDeviceSynchronize() != hipSuccess) return fail(AMPNET_E_LAUNCH, "ampnet_profile_read: device synchronize failed");
    int n = 0;
    for (auto &e : g_prof) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, e.e0, e.e1) != hipSuccess) continue;
        int k = 0;
        for (; k < n; ++k)
            if (strncmp(names + 64 * k, e.name, 64) == 0) break;
        if (k == n) {
            if (n >= max_rows) continue;
            memset(names + 64 * k, 0, 64);
            snprintf(names + 64 * k, 64, "%s", e.name);
            ms[k] = 0.0;
            calls[k] = 0;
            flops[k] = 0.0;
            bytes[k] = 0.0;
            ++n;
        }
        ms[k] += (double)t;
        calls[k] += 1;
        flops[k] += e.flops;
        bytes[k] += e.bytes;
    }
    return n;
}

namespace ampnet {
static int g_matrix_precision = AMPNET_PRECISION_F32;
int matrix_precision() { return g_matrix_precision; }
}  // namespace ampnet

// ---- forward-workspace tags: which precision mode a train-mode forward ran in ------------------------------------------------
// A backward entry point reads the forward's tape (the z tensors: fp32, or bf16 in mode 3) from the caller's workspace; the mode is
// a process-wide switch that may change between the two calls (another model, a test fixture, autograd replay).  The forward records
// (workspace base -> mode) on the host, the backward refuses a workspace whose forward ran in another mode instead of reinterpreting it.
namespace ampnet {
namespace {
std::mutex g_tag_mu;
struct WsTag {
    int mode;
    unsigned long long serial;       // when the forward ran (for eviction)
};
std::unordered_map<const void *, WsTag> g_ws_tag;
unsigned long long g_tag_serial = 0;
}  // namespace
void ws_tag_set(const void *ws, int mode)
{
    std::lock_guard<std::mutex> lk(g_tag_mu);
    if (g_ws_tag.size() >= 4096 && g_ws_tag.find(ws) == g_ws_tag.end()) {
        // workspaces come and go with the caller's allocator: drop the OLDER half (never a wholesale clear -- the encoder's tag of this very
        // step must survive the head's forward a moment later, or the backward would refuse a valid tape)
        std::vector<unsigned long long> serials;
        serials.reserve(g_ws_tag.size());
        for (const auto &kv : g_ws_tag) serials.push_back(kv.second.serial);
        std::nth_element(serials.begin(), serials.begin() + serials.size() / 2, serials.end());
        const unsigned long long cut = serials[serials.size() / 2];
        for (auto it = g_ws_tag.begin(); it != g_ws_tag.end();) it = it->second.serial < cut ? g_ws_tag.erase(it) : std::next(it);
    }
    g_ws_tag[ws] = WsTag{mode, ++g_tag_serial};
}
int ws_tag_check(const void *ws, const char *who)
{
    int have = -1;
    {
        std::lock_guard<std::mutex> lk(g_tag_mu);
        auto it = g_ws_tag.find(ws);
        if (it != g_ws_tag.end()) have = it->second.mode;
    }
    if (have < 0) return fail(AMPNET_E_ARG, "%s: this forward workspace holds no train-mode forward of this process", who);
    // modes 0..2 keep the saved activations in fp32 (a backward in any of them may follow a forward in any of them: the operand rounding
    // of the kernels differs, the tape does not); mode 3 keeps them in bf16
    const int now = matrix_precision();
    if ((have == AMPNET_PRECISION_BF16_STORE) != (now == AMPNET_PRECISION_BF16_STORE))
        return fail(AMPNET_E_ARG, "%s: the forward ran in matrix precision mode %d, the backward is called in mode %d "
                                  "(ampnet_set_matrix_precision changed in between): the saved activations (%s) would be misread", who, have, now,
                    have == AMPNET_PRECISION_BF16_STORE ? "bf16" : "fp32");
    return AMPNET_OK;
}
}  // namespace ampnet

extern "C" int ampnet_set_matrix_precision(int mode)
{
    if (mode != AMPNET_PRECISION_F32 && mode != AMPNET_PRECISION_BF16 && mode != AMPNET_PRECISION_BF16_TRAIN && mode != AMPNET_PRECISION_BF16_STORE &&
        mode != AMPNET_PRECISION_F32_SPLIT)
        return ampnet::fail(AMPNET_E_ARG, "ampnet_set_matrix_precision: mode %d", mode);
    ampnet::g_matrix_precision = mode;
    return AMPNET_OK;
}
extern "C" int ampnet_get_matrix_precision(void) { return ampnet::g_matrix_precision; }

extern "C" int ampnet_abi_version(void) { return AMPNET_ABI_VERSION; }
extern "C" const char *ampnet_last_error(void) { return ampnet::err_buf(); }

// ---- state_dict tables (ampnet_table_*): mirrors params.py of the Python package ----------------------
namespace {
struct Entry {
    const char *name;
    long numel;
};
#define TNET(pre, k)                                                                                                   \
    {pre "conv_1.weight", 64L * (k)}, {pre "conv_2.weight", 128L * 64}, {pre "conv_3.weight", 256L * 128},             \
    {pre "bn_1.weight", 64}, {pre "bn_1.bias", 64}, {pre "bn_2.weight", 128}, {pre "bn_2.bias", 128},                   \
    {pre "bn_3.weight", 256}, {pre "bn_3.bias", 256}, {pre "bn_4.weight", 256}, {pre "bn_4.bias", 256},                 \
    {pre "bn_5.weight", 128}, {pre "bn_5.bias", 128}, {pre "fc_1.weight", 256L * 256}, {pre "fc_2.weight", 128L * 256}, \
    {pre "fc_3.weight", 128L * (k) * (k)}, {pre "fc_3.bias", 1L * (k) * (k)}
#define BNBUF(pre, c) {pre "running_mean", c}, {pre "running_var", c}
const Entry kEncParams[] = {
    TNET("input_transform.", 3), TNET("feature_transform.", 64),
    {"conv_1.weight", 64L * 12}, {"conv_2.weight", 64L * 64}, {"conv_3.weight", 64L * 64}, {"conv_4.weight", 128L * 64},
    {"conv_5.weight", 128L * 128}, {"conv_6.weight", 256L * 128},
    {"bn_1.weight", 64}, {"bn_1.bias", 64}, {"bn_2.weight", 64}, {"bn_2.bias", 64}, {"bn_3.weight", 64}, {"bn_3.bias", 64},
    {"bn_4.weight", 128}, {"bn_4.bias", 128}, {"bn_5.weight", 128}, {"bn_5.bias", 128}, {"bn_6.weight", 256}, {"bn_6.bias", 256}};
const Entry kEncBuffers[] = {
    BNBUF("input_transform.bn_1.", 64), BNBUF("input_transform.bn_2.", 128), BNBUF("input_transform.bn_3.", 256),
    BNBUF("input_transform.bn_4.", 256), BNBUF("input_transform.bn_5.", 128),
    BNBUF("feature_transform.bn_1.", 64), BNBUF("feature_transform.bn_2.", 128), BNBUF("feature_transform.bn_3.", 256),
    BNBUF("feature_transform.bn_4.", 256), BNBUF("feature_transform.bn_5.", 128),
    BNBUF("bn_1.", 64), BNBUF("bn_2.", 64), BNBUF("bn_3.", 64), BNBUF("bn_4.", 128), BNBUF("bn_5.", 128), BNBUF("bn_6.", 256)};
const Entry kHeadParams[] = {
    {"fc1.weight", 32}, {"fc1.bias", 16}, {"fc2.weight", 256L * 16}, {"fc2.bias", 256},
    {"attention.in_proj_weight", 768L * 256}, {"attention.in_proj_bias", 768},
    {"attention.out_proj.weight", 256L * 256}, {"attention.out_proj.bias", 256},
    {"conv_2.weight", 128L * 320}, {"conv_2.bias", 128}, {"conv_3.weight", 64L * 128}, {"conv_3.bias", 64},
    {"conv_4.weight", 5L * 64}, {"conv_4.bias", 5},
    {"bn_2.weight", 128}, {"bn_2.bias", 128}, {"bn_3.weight", 64}, {"bn_3.bias", 64}};
const Entry kHeadBuffers[] = {BNBUF("bn_2.", 128), BNBUF("bn_3.", 64)};

const Entry *table(int t, int *n)
{
    switch (t) {
    case 0: *n = sizeof(kEncParams) / sizeof(Entry); return kEncParams;
    case 1: *n = sizeof(kEncBuffers) / sizeof(Entry); return kEncBuffers;
    case 2: *n = sizeof(kHeadParams) / sizeof(Entry); return kHeadParams;
    case 3: *n = sizeof(kHeadBuffers) / sizeof(Entry); return kHeadBuffers;
    default: *n = 0; return nullptr;
    }
}
}  // namespace

extern "C" int ampnet_table_count(int t)
{
    int n;
    table(t, &n);
    return n;
}
extern "C" const char *ampnet_table_name(int t, int i)
{
    int n;
    const Entry *e = table(t, &n);
    return (e && i >= 0 && i < n) ? e[i].name : nullptr;
}
extern "C" long ampnet_table_numel(int t, int i)
{
    int n;
    const Entry *e = table(t, &n);
    return (e && i >= 0 && i < n) ? e[i].numel : -1;
}

// core.hip -- error plumbing and version entry points of the C ABI (include/ampnet_hip.h).
#include "common.h"

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

}  // namespace ampnet

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

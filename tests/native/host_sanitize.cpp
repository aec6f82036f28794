// Host-side logic of the C-ABI library under AddressSanitizer + UndefinedBehaviorSanitizer, WITHOUT a GPU: the library's sources
// are compiled host-only (hipcc --offload-host-only -fsanitize=address,undefined; `make -C vidsgg-big_amd/csrc asan`) and linked
// with this driver.  It walks what runs before the first kernel launch: argument checks of bigc_create, the parameter table
// (bigc_set_param), the resolver of bigc_finalize (missing parameter / wrong element count -> error code + message, no HIP call
// reached), the workspace arena layout (bigc_workspace_bytes: pure size arithmetic over the config and the batch), the mode
// switches, the profiling table, and bigc_destroy.  Device pointers are made-up 16-byte-aligned addresses: nothing here
// dereferences them.  Input: a spec file written by tests/test_host_cpu.py:
//     model <17 config ints>
//     param <name> <numel>          (every state_dict key of that model)
//     batch <n_vid> <n_trk> <n_rows> <n_conv> <max_trk>
#include "../../include/bigc_hip.h"
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#define REQUIRE(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s -- ", __FILE__, __LINE__, #cond); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } } while (0)

struct Param { std::string name; long long numel; };
struct Batch { int v[5]; };
struct Spec { bigc_config cfg; std::vector<Param> params; std::vector<Batch> batches; };

static const float* fake_ptr(size_t i) { return (const float*)(uintptr_t)(0x10000000ull + i * 0x100000ull); }

static void run_model(const Spec& sp) {
    bigc_model* m = nullptr;
    REQUIRE(bigc_create(&sp.cfg, &m) == BIGC_OK && m, "create: %s", bigc_last_error());
    // argument checks
    REQUIRE(bigc_set_param(m, "x", nullptr, 4) != BIGC_OK, "null pointer accepted");
    REQUIRE(bigc_set_param(m, "x", (const float*)(uintptr_t)0x10000004ull, 4) != BIGC_OK, "misaligned pointer accepted");
    REQUIRE(strstr(bigc_last_error(), "aligned") != nullptr, "message: %s", bigc_last_error());
    // nothing set: finalize names the first missing parameter and reaches no HIP call
    int rc = bigc_finalize(m, nullptr);
    REQUIRE(rc == BIGC_ERR_MISSING, "finalize of an empty model: rc %d (%s)", rc, bigc_last_error());
    for (size_t i = 0; i < sp.params.size(); ++i)
        REQUIRE(bigc_set_param(m, sp.params[i].name.c_str(), fake_ptr(i), sp.params[i].numel) == BIGC_OK, "set_param %s: %s", sp.params[i].name.c_str(), bigc_last_error());
    // one parameter with a wrong element count -> BIGC_ERR_ARG naming it (still before any HIP call)
    const size_t victim = sp.params.size() / 2;
    REQUIRE(bigc_set_param(m, sp.params[victim].name.c_str(), fake_ptr(victim), sp.params[victim].numel + 4) == BIGC_OK, "set_param");
    rc = bigc_finalize(m, nullptr);
    const bool named = strstr(bigc_last_error(), sp.params[victim].name.c_str()) != nullptr;
    // (a key the model does not read -- e.g. a buffer of the state_dict -- is simply ignored: then finalize goes on to the GPU part)
    REQUIRE((rc == BIGC_ERR_ARG && named) || rc == BIGC_ERR_HIP || rc == BIGC_OK, "wrong numel: rc %d (%s)", rc, bigc_last_error());
    REQUIRE(bigc_set_param(m, sp.params[victim].name.c_str(), fake_ptr(victim), sp.params[victim].numel) == BIGC_OK, "set_param");
    // re-registering the same key replaces it (no growth of the table)
    for (int rep = 0; rep < 3; ++rep)
        for (size_t i = 0; i < sp.params.size(); ++i) REQUIRE(bigc_set_param(m, sp.params[i].name.c_str(), fake_ptr(i + rep), sp.params[i].numel) == BIGC_OK, "set_param");
    // workspace layout: positive, 256-byte granular, monotone in the batch size, no overflow for the largest batch
    size_t prev = 0;
    for (const Batch& bt : sp.batches) {
        bigc_batch b;
        memset(&b, 0, sizeof(b));
        b.n_vid = bt.v[0]; b.n_trk = bt.v[1]; b.n_rows = bt.v[2]; b.n_conv = bt.v[3]; b.max_trk = bt.v[4];
        const size_t bytes = bigc_workspace_bytes(m, &b);
        REQUIRE(bytes > 0 && bytes % 256 == 0, "workspace_bytes = %zu", bytes);
        REQUIRE(bytes >= prev, "workspace not monotone: %zu after %zu", bytes, prev);
        // at least the three largest tensors of the forward must fit (rows x 2 dim_enti, conv rows x dim_enti, B Q x 2 dim_pred)
        const size_t lower = 4 * ((size_t)b.n_rows * 2 * sp.cfg.dim_enti + (size_t)b.n_conv * sp.cfg.dim_enti + (size_t)b.n_vid * sp.cfg.num_querys * 2 * sp.cfg.dim_pred);
        REQUIRE(bytes >= lower, "workspace %zu smaller than its largest tensors %zu", bytes, lower);
        prev = bytes;
        printf("  batch %d videos / %d tracklets / %d rows: %.1f MB\n", b.n_vid, b.n_trk, b.n_rows, bytes / 1e6);
    }
    REQUIRE(bigc_workspace_bytes(nullptr, nullptr) == 0, "null model");
    bigc_destroy(m);
    bigc_destroy(nullptr);
}

int main(int argc, char** argv) {
    REQUIRE(argc == 2, "usage: host_sanitize <spec file>");
    std::ifstream in(argv[1]);
    REQUIRE((bool)in, "cannot open %s", argv[1]);
    std::vector<Spec> specs;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string kind;
        ss >> kind;
        if (kind == "model") {
            specs.emplace_back();
            int32_t* f = (int32_t*)&specs.back().cfg;
            for (size_t i = 0; i < sizeof(bigc_config) / sizeof(int32_t); ++i) REQUIRE((bool)(ss >> f[i]), "model line: %zu config fields expected", sizeof(bigc_config) / sizeof(int32_t));
        } else if (kind == "param") {
            Param p;
            ss >> p.name >> p.numel;
            REQUIRE(!specs.empty() && p.numel > 0, "bad param line '%s'", line.c_str());
            specs.back().params.push_back(p);
        } else if (kind == "batch") {
            Batch b;
            for (int i = 0; i < 5; ++i) ss >> b.v[i];
            REQUIRE(!specs.empty(), "batch before model");
            specs.back().batches.push_back(b);
        }
    }
    REQUIRE(!specs.empty(), "empty spec");
    // bad configurations are refused with BIGC_ERR_ARG and a message
    {
        bigc_config c = specs[0].cfg;
        bigc_model* m = nullptr;
        c.n_att_head = 7;
        REQUIRE(bigc_create(&c, &m) == BIGC_ERR_ARG && strlen(bigc_last_error()) > 0, "heads not dividing the width accepted");
        c = specs[0].cfg; c.variant = 9;
        REQUIRE(bigc_create(&c, &m) == BIGC_ERR_ARG, "variant 9 accepted");
        c = specs[0].cfg; c.num_pred_cats = 1000;
        REQUIRE(bigc_create(&c, &m) == BIGC_ERR_ARG, "1000 categories accepted");
        REQUIRE(bigc_create(nullptr, &m) == BIGC_ERR_ARG && bigc_create(&specs[0].cfg, nullptr) == BIGC_ERR_ARG, "null pointers accepted");
    }
    for (size_t i = 0; i < specs.size(); ++i) {
        printf("model %zu: variant %d, %zu parameters\n", i, specs[i].cfg.variant, specs[i].params.size());
        run_model(specs[i]);
    }
    // mode switches return the previous mode; the exact mode masks the fast one
    const int e0 = bigc_linear_set_exact(0), f0 = bigc_linear_set_fast(1);
    REQUIRE(bigc_linear_get_fast() == 1 && bigc_linear_set_exact(1) == 0 && bigc_linear_get_fast() == 0, "mode switches");
    bigc_linear_set_exact(e0); bigc_linear_set_fast(f0);
    // profiling table: names for every family, out-of-range queries refused
    const int nf = bigc_prof_family_count();
    REQUIRE(nf > 0, "no profiling families");
    for (int f = 0; f < nf; ++f) REQUIRE(strlen(bigc_prof_family_name(f)) > 0, "family %d has no name", f);
    double ms, fl, by; long long n;
    REQUIRE(bigc_prof_get(nf, &ms, &n, &fl, &by) != BIGC_OK && bigc_prof_get(-1, &ms, &n, &fl, &by) != BIGC_OK, "out-of-range family accepted");
    REQUIRE(strstr(bigc_build_info(), "gfx950") != nullptr, "build info: %s", bigc_build_info());
    printf("host_sanitize: OK\n");
    return 0;
}

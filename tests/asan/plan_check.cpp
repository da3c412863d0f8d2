// CPU-only exercise of the host side of libcae_hip under AddressSanitizer: plan creation / tensor tables / error paths /
// destruction of the ConvAE engine (plain and trunk mode through the var engine) and the UNET engine.  No GPU call is made.
#include <cstdio>
#include <cstring>
#include <vector>
#include "cae_hip.h"
#include "cae_vae.h"
#include "cae_unet.h"

static std::vector<cae_layer_spec> enc_layers(int h, int n, int c0) {
    std::vector<cae_layer_spec> v;
    int c = c0, s = h;
    for (int i = 0; i < n; i++) {
        cae_layer_spec l{c, s, s, c * 2, (s - 3) / 2 + 1, (s - 3) / 2 + 1, 3, 3, 2, 0};
        v.push_back(l);
        c *= 2;
        s = (s - 3) / 2 + 1;
    }
    return v;
}
static std::vector<cae_layer_spec> dec_layers(int h, int n, int c0, int last_k) {
    std::vector<cae_layer_spec> v;
    int c = c0, s = h;
    for (int i = 0; i < n; i++) {
        const int k = i == n - 1 ? last_k : 3;
        const int co = c / 2 > 0 ? c / 2 : 1;
        cae_layer_spec l{c, s, s, co, (s - 1) * 2 + k, (s - 1) * 2 + k, k, k, 2, 0};
        v.push_back(l);
        c = co;
        s = (s - 1) * 2 + k;
    }
    return v;
}

int main() {
    int failures = 0;
    auto expect = [&](bool ok, const char* what) {
        if (!ok) {
            printf("FAILED: %s (%s)\n", what, cae_last_error());
            failures++;
        }
    };
    for (int rep = 0; rep < 3; rep++) {
        auto enc = enc_layers(16, 2, 1);
        auto dec = dec_layers(3, 6, 64, 4);
        cae_engine* e = nullptr;
        expect(cae_engine_create(enc.data(), (int)enc.size(), dec.data(), (int)dec.size(), 128, 32, 64, &e) == 0, "cae_engine_create");
        if (e) {
            cae_tensor_info_t t;
            long long total = 0;
            for (int i = 0; i < cae_tensor_count(e); i++) {
                expect(cae_tensor_info(e, i, &t) == 0, "cae_tensor_info");
                if (t.arena == 0) total += t.numel;
            }
            expect(total == 112271, "parameter count of the benchmark geometry");
            expect(cae_tensor_info(e, cae_tensor_count(e), &t) != 0, "tensor index out of range is refused");
            expect(cae_workspace_bytes(e) > 0 && cae_param_count(e) >= total, "sizes");
            expect(cae_train_step(e, 0, nullptr, 64) != 0, "a step on an unbound engine is refused");
            expect(cae_set_cursor(e, 0, 0) != 0, "cursor on an unbound engine is refused");
            cae_engine_destroy(e);
        }
        // broken geometry: messages, no leak
        auto bad = dec;
        bad[2].out_h += 1;
        cae_engine* b = nullptr;
        expect(cae_engine_create(enc.data(), 2, bad.data(), 6, 128, 32, 64, &b) != 0 && b == nullptr, "inconsistent decoder is refused");
        expect(cae_engine_create(nullptr, 0, dec.data(), 6, 128, 32, 64, &b) != 0, "null encoder is refused");
        // the 'var' engine = a trunk-mode ConvAE engine inside
        auto venc = enc_layers(64, 4, 1);
        auto vdec = dec_layers(3, 7, 128, 4);
        vae_engine* v = nullptr;
        expect(vae_engine_create(venc.data(), 4, vdec.data(), 7, 128, 32, 16, &v) == 0, "vae_engine_create");
        if (v) {
            cae_tensor_info_t t;
            bool mu = false, lv = false;
            for (int i = 0; i < vae_tensor_count(v); i++) {
                expect(vae_tensor_info(v, i, &t) == 0, "vae_tensor_info");
                mu |= !strcmp(t.name, "enc/encoder_mu.weight");
                lv |= !strcmp(t.name, "enc/encoder_logvar.bias");
            }
            expect(mu && lv, "the heads are listed under the var model's names");
            expect(vae_train_step(v, 0, nullptr, 0, 16, 0) != 0, "a step on an unbound var engine is refused");
            vae_engine_destroy(v);
        }
        auto small = dec_layers(3, 5, 64, 4);   // 128x128 output: MS-SSIM needs >= 176
        expect(vae_engine_create(enc.data(), 2, small.data(), 5, 16, 4, 4, &v) != 0, "too small an output for MS-SSIM is refused");
    }
    printf(failures ? "%d checks failed\n" : "host-side plan checks clean (%d)\n", failures);
    return failures ? 1 : 0;
}

// tests/c_abi/c_abi_parity.cpp -- TEST INFRASTRUCTURE.  A C++ trainer's view of the drop-in boundary: no Python, no torch.
// It drives libmemgym_hip.so through include/memgym.h with hipMalloc'd buffers on its own stream and checks every frame,
// reward and done flag bit-exactly against the CPU oracle (oracle/_build/libmemgym_oracle.so, the checker only).
// Usage: c_abi_parity ENV_ID NUM_ENVS STEPS   -> prints "OK ..." and exits 0, or the first mismatch and exits 1.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "memgym.h"

extern "C" {  // oracle/mgo_api.c
struct mgo_batch;
mgo_batch* mgo_batch_create(const char* env_id, int n, double scale);
void mgo_batch_destroy(mgo_batch* b);
void mgo_batch_reset(mgo_batch* b, const int64_t* seeds, uint8_t* obs);
void mgo_batch_step(mgo_batch* b, const int32_t* actions, int autoreset, uint8_t* obs, double* reward, uint8_t* done);
}

#define HIP_OK(e)                                                                  \
    do {                                                                           \
        hipError_t e_ = (e);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_));                \
            return 2;                                                              \
        }                                                                          \
    } while (0)
#define MG_OK(e)                                                                   \
    do {                                                                           \
        if ((e) != 0) {                                                            \
            fprintf(stderr, "%s: %s\n", #e, mg_last_error());                      \
            return 2;                                                              \
        }                                                                          \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s ENV_ID NUM_ENVS STEPS\n", argv[0]);
        return 2;
    }
    const char* env_id = argv[1];
    const int n = atoi(argv[2]), steps = atoi(argv[3]);
    const size_t frame = 84 * 84 * 3;

    HIP_OK(hipSetDevice(0));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    mg_env* env = nullptr;
    MG_OK(mg_create(env_id, n, 0, &env));
    const int adim = mg_action_dim(env), n_act = adim == 1 ? 4 : 3;

    // capacities of the lists the reference grows without limit (mg_set_capacity, before the first reset): a larger store changes where an
    // instance's records lie, nothing else -- the run below must equal the oracle all the same
    const bool emp = !strcmp(env_id, "Endless-MysteryPath-v0"), emm = !strcmp(env_id, "Endless-MortarMayhem-v0");
    if (emp) {
        if (mg_capacity(env, "path_segments") != 128 || mg_capacity(env, "fall_off_cells") != 128) { printf("default capacities differ from the header's\n"); return 1; }
        MG_OK(mg_set_capacity(env, "path_segments", 200));
        if (mg_capacity(env, "path_segments") != 200) { printf("mg_capacity does not return what mg_set_capacity set\n"); return 1; }
        if (mg_set_capacity(env, "path_segments", 2) == 0 || mg_set_capacity(env, "commands", 600) == 0) { printf("a bad capacity was accepted\n"); return 1; }
    } else if (emm) {
        if (mg_capacity(env, "commands") != 512) { printf("default command capacity differs from the header's\n"); return 1; }
        MG_OK(mg_set_capacity(env, "commands", 700));
    } else if (mg_capacity(env, "path_segments") != -1 || mg_set_capacity(env, "path_segments", 200) == 0) {
        printf("an env id without that list accepted its capacity\n");
        return 1;
    }

    uint8_t *obs_d, *done_d;
    float* rew_d;
    int32_t* act_d;
    int64_t* seeds_d;
    HIP_OK(hipMalloc((void**)&obs_d, frame * n));
    HIP_OK(hipMalloc((void**)&done_d, n));
    HIP_OK(hipMalloc((void**)&rew_d, sizeof(float) * n));
    HIP_OK(hipMalloc((void**)&act_d, sizeof(int32_t) * n * adim));
    HIP_OK(hipMalloc((void**)&seeds_d, sizeof(int64_t) * n));

    std::vector<int64_t> seeds(n);
    for (int i = 0; i < n; ++i) seeds[i] = 100 + i;
    HIP_OK(hipMemcpy(seeds_d, seeds.data(), sizeof(int64_t) * n, hipMemcpyHostToDevice));

    mgo_batch* ref = mgo_batch_create(env_id, n, 0.25);
    if (!ref) {
        fprintf(stderr, "oracle: unknown env id %s\n", env_id);
        return 2;
    }
    std::vector<uint8_t> obs(frame * n), want(frame * n), done(n), want_done(n);
    std::vector<float> rew(n);
    std::vector<double> want_rew(n);
    std::vector<int32_t> act((size_t)n * adim);

    MG_OK(mg_reset(env, seeds_d, nullptr, obs_d, nullptr, stream));
    HIP_OK(hipMemcpyAsync(obs.data(), obs_d, frame * n, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    mgo_batch_reset(ref, seeds.data(), want.data());
    if (memcmp(obs.data(), want.data(), frame * n) != 0) {
        printf("MISMATCH in the reset frames\n");
        return 1;
    }
    // the versioned info struct: the step reward as the reference's Python float (a double) next to its float32 rounding
    double* rew64_d;
    HIP_OK(hipMalloc((void**)&rew64_d, sizeof(double) * n));
    std::vector<double> rew64(n);
    mg_info_buffers info;
    memset(&info, 0, sizeof(info));
    info.reward64_dev = rew64_d;
    uint8_t* cap_d;  // (round 6) 1 where an episode was ended on a capacity of the build: never in this run
    HIP_OK(hipMalloc((void**)&cap_d, n));
    HIP_OK(hipMemset(cap_d, 0xFF, n));
    info.capacity_dev = cap_d;
    std::vector<uint8_t> cap(n);
    if (mg_step(env, act_d, obs_d, rew_d, done_d, nullptr, &info, 1, stream) == 0) {  // struct_size still 0: must be refused
        printf("a mg_info_buffers without struct_size was accepted\n");
        return 1;
    }
    info.struct_size = (size_t)(uintptr_t)rew64_d;  // what a caller built against the round-2 header (no struct_size) has here: a pointer
    if (mg_step(env, act_d, obs_d, rew_d, done_d, nullptr, &info, 1, stream) == 0) {
        printf("a device pointer in the place of struct_size was accepted\n");
        return 1;
    }
    info.struct_size = sizeof(info);
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
    long episodes = 0;
    for (int t = 0; t < steps; ++t) {
        for (size_t k = 0; k < act.size(); ++k) {
            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
            act[k] = (int32_t)((lcg >> 33) % (uint64_t)n_act);
        }
        HIP_OK(hipMemcpyAsync(act_d, act.data(), sizeof(int32_t) * act.size(), hipMemcpyHostToDevice, stream));
        MG_OK(mg_step(env, act_d, obs_d, rew_d, done_d, nullptr, &info, 1, stream));
        HIP_OK(hipMemcpyAsync(rew64.data(), rew64_d, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(obs.data(), obs_d, frame * n, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(rew.data(), rew_d, sizeof(float) * n, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(done.data(), done_d, n, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(cap.data(), cap_d, n, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        mgo_batch_step(ref, act.data(), 1, want.data(), want_rew.data(), want_done.data());
        for (int i = 0; i < n; ++i) {
            if (cap[i] != 0) {
                printf("capacity_dev[%d] = %d at step %d (written by every step, 0 unless the episode ended on a capacity)\n", i, cap[i], t);
                return 1;
            }
            if (done[i] != want_done[i] || rew[i] != (float)want_rew[i] || rew64[i] != want_rew[i] || memcmp(&obs[frame * i], &want[frame * i], frame) != 0) {
                printf("MISMATCH at step %d, instance %d: done %d/%d reward %g/%g\n", t, i, done[i], want_done[i], rew[i], want_rew[i]);
                return 1;
            }
            episodes += done[i];
        }
    }
    int flags = 0;
    MG_OK(mg_poll_errors(env, &flags));
    if (flags) {
        printf("device error flags 0x%x\n", flags);
        return 1;
    }
    {   // checkpoint blob: header checked, a blob of another handle shape refused
        std::vector<char> blob(mg_state_size(env));
        MG_OK(mg_get_state(env, blob.data(), blob.size()));
        MG_OK(mg_set_state(env, blob.data(), blob.size()));
        mg_env* other = nullptr;
        MG_OK(mg_create(env_id, n + 1, 0, &other));
        if (mg_set_capacity(env, "path_segments", 300) == 0) {
            printf("mg_set_capacity after the first reset was accepted\n");
            return 1;
        }
        if (mg_set_state(other, blob.data(), blob.size()) == 0) {
            printf("a state blob of %d instances was accepted by a handle of %d\n", n, n + 1);
            return 1;
        }
        mg_destroy(other);
        blob[0] ^= 1;
        if (mg_set_state(env, blob.data(), blob.size()) == 0) {
            printf("a state blob with a broken magic was accepted\n");
            return 1;
        }
    }
    {   // (round 6) a float output format into a buffer from mg_obs_alloc_for: two 304-MiB pieces told that an observation is 42,336 bytes,
        // the n observations laid ACROSS the boundary between the pieces; bfloat16 = the float32 quotient byte / 255 rounded to nearest even
        const size_t fb = frame * 2, piece = (size_t)304 << 20, bytes = 2 * piece;
        int zones[4] = {-1, -1, -1, -1};
        size_t pb = 0, lead = 0;
        if (mg_obs_plan(bytes, fb, 2, &pb, &lead, zones, 4) != 2 || pb != piece || lead != 0 || zones[0] == zones[1]) {
            printf("mg_obs_plan: two pieces of one window must come from different zones (%d %d, piece %zu, lead %zu)\n", zones[0], zones[1], pb, lead);
            return 1;
        }
        void* big = nullptr;
        mg_obs_alloc_info ai;
        MG_OK(mg_obs_alloc_for(0, bytes, fb, MG_OBS_SEARCH_DEFAULT, &big, &ai));
        uint8_t* at = static_cast<uint8_t*>(big) + piece - (size_t)(n / 2) * fb;
        MG_OK(mg_set_obs_format(env, MG_OBS_BF16_CYX));
        if (mg_obs_bytes(env) != fb) { printf("mg_obs_bytes: %zu for a bfloat16 observation\n", mg_obs_bytes(env)); return 1; }
        MG_OK(mg_render(env, at, stream));
        std::vector<uint16_t> got((size_t)frame * n);
        HIP_OK(hipMemcpyAsync(got.data(), at, fb * n, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < 3; ++c)
                for (int y = 0; y < 84; ++y)
                    for (int x = 0; x < 84; ++x) {
                        const float q = (float)want[frame * i + ((size_t)x * 84 + y) * 3 + c] / 255.0f;  // the oracle's frame is [x][y][c]
                        uint32_t b;
                        memcpy(&b, &q, 4);
                        const uint16_t w = (uint16_t)((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16);
                        if (got[frame * i + ((size_t)c * 84 + y) * 84 + x] != w) {
                            printf("MISMATCH in the bfloat16 observation of instance %d at (c %d, y %d, x %d)\n", i, c, y, x);
                            return 1;
                        }
                    }
        MG_OK(mg_set_obs_format(env, MG_OBS_U8_XYC));
        MG_OK(mg_obs_free(big));
    }
    printf("OK %s: %d instances x %d steps, %ld episodes finished, bit-exact through the C ABI\n", env_id, n, steps, episodes);
    (void)hipFree(rew64_d);
    (void)hipFree(cap_d);
    mg_destroy(env);
    mgo_batch_destroy(ref);
    (void)hipFree(obs_d); (void)hipFree(done_d); (void)hipFree(rew_d); (void)hipFree(act_d); (void)hipFree(seeds_d);
    (void)hipStreamDestroy(stream);
    return 0;
}

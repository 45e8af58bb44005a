// packinfo.cpp -- host-side builder of the sequence-packing index arrays (K9).
// Replaces the numpy routine build_pack_info_from_dones / build_pack_info_from_episode_ids
// (rl/models/rnn_state_encoder.py:35-168).  Same definition of the arrays; ties between fragments
// of equal length are broken by (episode id, env) order (a stable sort), whereas numpy's default
// argsort leaves tie order to its sort implementation -- any tie order yields the same RNN result.
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "../../include/habitat_amd.h"

extern "C" int hab_build_pack_info(const uint8_t* dones, int T, int N, int64_t* select_inds, int64_t* num_seqs_at_step,
                                   int64_t* sequence_starts, int64_t* sequence_lengths, int64_t* rnn_state_batch_inds,
                                   uint8_t* last_sequence_in_batch_mask, uint8_t* first_sequence_in_batch_mask,
                                   int64_t* last_sequence_in_batch_inds, int64_t* first_episode_in_batch_inds,
                                   int64_t* first_step_for_env, int32_t* num_fragments, int32_t* max_len) {
    if (!dones || T <= 0 || N <= 0 || !select_inds || !num_seqs_at_step || !sequence_starts || !sequence_lengths ||
        !rnn_state_batch_inds || !num_fragments || !max_len)
        return -1;
    struct Frag { int64_t ep; int env; int t0; int len; };
    std::vector<Frag> frags;
    frags.reserve((size_t)N * 4);
    // fragments of env n: start at t = 0 and at every t > 0 with a done; episode id = inclusive cumsum of dones
    for (int n = 0; n < N; ++n) {
        int64_t ep = dones[n] != 0 ? 1 : 0;  // id of the fragment that starts at t = 0
        int start = 0;
        for (int t = 1; t < T; ++t) {
            if (dones[(size_t)t * N + n] != 0) {
                frags.push_back({ep, n, start, t - start});
                ++ep;
                start = t;
            }
        }
        frags.push_back({ep, n, start, T - start});
    }
    std::sort(frags.begin(), frags.end(), [](const Frag& a, const Frag& b) { return a.ep != b.ep ? a.ep < b.ep : a.env < b.env; });
    std::stable_sort(frags.begin(), frags.end(), [](const Frag& a, const Frag& b) { return a.len > b.len; });
    const int F = (int)frags.size();
    const int L = frags[0].len;
    *num_fragments = F;
    *max_len = L;
    int64_t p = 0;
    int active = F;
    for (int s = 0; s < L; ++s) {
        while (active > 0 && frags[active - 1].len <= s) --active;
        num_seqs_at_step[s] = active;
        for (int q = 0; q < active; ++q) select_inds[p++] = (int64_t)(frags[q].t0 + s) * N + frags[q].env;
    }
    std::vector<int64_t> env_min(N, INT64_MAX), env_max(N, -1);
    for (int q = 0; q < F; ++q) {
        sequence_starts[q] = (int64_t)frags[q].t0 * N + frags[q].env;
        sequence_lengths[q] = frags[q].len;
        rnn_state_batch_inds[q] = frags[q].env;
        env_min[frags[q].env] = std::min(env_min[frags[q].env], frags[q].ep);
        env_max[frags[q].env] = std::max(env_max[frags[q].env], frags[q].ep);
    }
    int nl = 0, nf = 0;
    for (int q = 0; q < F; ++q) {
        const bool last = frags[q].ep == env_max[frags[q].env];
        const bool first = frags[q].ep == env_min[frags[q].env];
        if (last_sequence_in_batch_mask) last_sequence_in_batch_mask[q] = last;
        if (first_sequence_in_batch_mask) first_sequence_in_batch_mask[q] = first;
        if (last && last_sequence_in_batch_inds) last_sequence_in_batch_inds[nl++] = q;
        if (first && first_episode_in_batch_inds) first_episode_in_batch_inds[nf++] = q;
        if (first && first_step_for_env) first_step_for_env[frags[q].env] = sequence_starts[q];
    }
    return 0;
}

// General form (build_pack_info_from_episode_ids, rnn_state_encoder.py:35-150): P frames in ANY order, each tagged with
// (episode id, environment id, step id) -- what a VER minibatch is (rl/ver/ver_rollout_storage.py:586-617).  A fragment is the set
// of frames of one (environment, episode), ordered by step id.  Environments are renumbered 0..n-1 in increasing id order
// (np.unique), rnn_state_batch_inds refers to that numbering and first_step_for_env[e] is the first frame of environment e's
// lowest-numbered episode: the frame whose stored hidden state seeds EVERY fragment of that environment (masked to zero where the
// fragment's first frame is an episode start), exactly like build_rnn_inputs does (:232-239).
extern "C" int hab_build_pack_info_from_ids(const int64_t* episode_ids, const int64_t* environment_ids, const int64_t* step_ids, int P,
                                            int64_t* select_inds, int64_t* num_seqs_at_step, int64_t* sequence_starts,
                                            int64_t* sequence_lengths, int64_t* rnn_state_batch_inds,
                                            uint8_t* last_sequence_in_batch_mask, uint8_t* first_sequence_in_batch_mask,
                                            int64_t* first_step_for_env, int32_t* num_fragments, int32_t* max_len, int32_t* num_envs) {
    if (!episode_ids || !environment_ids || !step_ids || P <= 0 || !select_inds || !num_seqs_at_step || !sequence_starts ||
        !sequence_lengths || !rnn_state_batch_inds || !num_fragments || !max_len || !num_envs)
        return -1;
    std::vector<int> order(P);
    for (int i = 0; i < P; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) {
        if (episode_ids[a] != episode_ids[b]) return episode_ids[a] < episode_ids[b];
        if (environment_ids[a] != environment_ids[b]) return environment_ids[a] < environment_ids[b];
        return step_ids[a] < step_ids[b];
    });
    struct Frag { int64_t ep, env; int first, len; };  // first: position in `order`
    std::vector<Frag> frags;
    for (int i = 0; i < P;) {
        int j = i + 1;
        while (j < P && episode_ids[order[j]] == episode_ids[order[i]] && environment_ids[order[j]] == environment_ids[order[i]]) {
            if (step_ids[order[j]] == step_ids[order[j - 1]]) return -1;  // duplicate (episode, env, step)
            ++j;
        }
        frags.push_back({episode_ids[order[i]], environment_ids[order[i]], i, j - i});
        i = j;
    }
    std::stable_sort(frags.begin(), frags.end(), [](const Frag& a, const Frag& b) { return a.len > b.len; });
    const int F = (int)frags.size(), L = frags[0].len;
    *num_fragments = F;
    *max_len = L;
    int64_t p = 0;
    int active = F;
    for (int s = 0; s < L; ++s) {
        while (active > 0 && frags[active - 1].len <= s) --active;
        num_seqs_at_step[s] = active;
        for (int q = 0; q < active; ++q) select_inds[p++] = order[frags[q].first + s];
    }
    std::vector<int64_t> envs;
    for (const Frag& f : frags) envs.push_back(f.env);
    std::sort(envs.begin(), envs.end());
    envs.erase(std::unique(envs.begin(), envs.end()), envs.end());
    const int n = (int)envs.size();
    *num_envs = n;
    std::vector<int64_t> env_min(n, INT64_MAX), env_max(n, INT64_MIN);
    for (int q = 0; q < F; ++q) {
        const int e = (int)(std::lower_bound(envs.begin(), envs.end(), frags[q].env) - envs.begin());
        sequence_starts[q] = order[frags[q].first];
        sequence_lengths[q] = frags[q].len;
        rnn_state_batch_inds[q] = e;
        env_min[e] = std::min(env_min[e], frags[q].ep);
        env_max[e] = std::max(env_max[e], frags[q].ep);
    }
    for (int q = 0; q < F; ++q) {
        const int e = (int)rnn_state_batch_inds[q];
        const bool last = frags[q].ep == env_max[e], first = frags[q].ep == env_min[e];
        if (last_sequence_in_batch_mask) last_sequence_in_batch_mask[q] = last;
        if (first_sequence_in_batch_mask) first_sequence_in_batch_mask[q] = first;
        if (first && first_step_for_env) first_step_for_env[e] = sequence_starts[q];
    }
    return 0;
}

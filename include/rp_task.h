/* rp_task.h — fused task-layer kernels for PianoWithShadowHands (C ABI, gfx950).
 *
 * The reference computes its rewards in Python/numpy once per control step
 * (robopianist/suite/tasks/piano_with_shadow_hands.py:251-331, summed by
 * suite/composite_reward.py:46-56).  Batched over environments these are ~130 tiny
 * elementwise launches; rp_task_rewards evaluates all terms for all envs in one launch
 * (one wavefront per env, lane = piano key).
 *
 * All pointers are DEVICE pointers into caller-owned memory; floating arrays have the
 * element type selected by `precision` (32: float, 64: double).  The launch is enqueued on
 * `hip_stream` (hipStream_t; NULL = default stream) and never synchronises.
 * Returns 0, or a negative code with the message in rp_task_last_error().
 */
#ifndef RP_TASK_H_
#define RP_TASK_H_

#ifdef __cplusplus
extern "C" {
#endif

#define RP_TASK_N_KEYS 88
#define RP_TASK_N_TERMS 5 /* key_press, sustain, energy, fingering, forearm */

typedef struct rp_task_reward_args {
  int n_envs, precision;
  int nv, nu, n_sites, n_contacts;   /* row lengths of qpos, act_*, site_xpos, contact_geoms */
  int use_fingering, use_forearm;    /* disabled terms are written as 0 and not summed.  use_fingering: 1 = the
                                      * MIDI's fingering (:300-331); 2 = the optimal-transport term (:333-369):
                                      * minimum-cost assignment between the fingertips and the keys to press
                                      * (scipy.optimize.linear_sum_assignment in the reference), term slot 3 */
  double energy_coef;                /* _ENERGY_PENALTY_COEF (:24) */
  double key_close, finger_close;    /* _KEY_CLOSE_ENOUGH_TO_PRESSED, _FINGER_CLOSE_ENOUGH_TO_KEY (:22-23) */
  /* engine state: rp_field_ptr views */
  const void* qpos;                  /* [E][nv] */
  const void* act_force;             /* [E][nu]   actuatorfrc sensors (shadow_hand.py:407-416) */
  const void* act_vel;               /* [E][nu]   actuatorvel sensors */
  const void* site_xpos;             /* [E][n_sites][3] */
  const int* contact_geoms;          /* [E][n_contacts][2], -1 = empty */
  /* task state */
  const void* goal_current;          /* [E][89]  goal keys + sustain (:196-198) */
  const void* key_norm_state;        /* [E][88]  Piano.normalized_state (piano.py:186-189) */
  const unsigned char* key_activation;      /* [E][88] */
  const unsigned char* sustain_activation;  /* [E] */
  const long long* finger_current;   /* [E][88]  finger index of each goal key, -1 = none */
  /* model constants */
  const int* key_qadr;               /* [88] qpos address of every key joint */
  const void* key_anchor;            /* [88][3] hinge position */
  const void* key_half;              /* [88][3] key box half sizes */
  const int* hand_act; int n_hand_act;      /* actuator ids of both hands (energy term) */
  const int* tip_site;               /* [10] engine site index of the fingertips, right hand first ([5] with hand_filter) */
  const int* rfa; int n_rfa;         /* right / left forearm geom ids (:251-259) */
  const int* lfa; int n_lfa;
  /* outputs */
  void* terms;                       /* [RP_TASK_N_TERMS][E] */
  void* total;                       /* [E] sum of the enabled terms in the order above */
  /* PianoWithOneShadowHand (piano_with_one_shadow_hand.py:237-311): 0 = two hands (finger ids 0-9,
   * tip_site[10], fingering_state [E][10]); 1 = right hand only, 2 = left hand only (tip_site[5],
   * fingering_state [E][5]; finger_current / finger_next hold the index into this hand's fingertips,
   * -1 for goal keys fingered by the other hand, and only this hand's keys enter the fingering term) */
  int hand_filter;
} rp_task_reward_args;

int rp_task_rewards(const rp_task_reward_args* args, void* hip_stream);

/* rp_task_advance: everything the reference does in Python between the last physics
 * substep and the TimeStep it returns, for all envs in one launch:
 *   Piano._update_key_state            (models/piano/piano.py:178-192)
 *   PianoWithShadowHands.after_step    (suite/tasks/piano_with_shadow_hands.py:188-204)
 *   goal / fingering observables       (:371-412)
 *   the reward terms                   (:251-331, as rp_task_rewards)
 *   should_terminate_episode, discount (:209-220)
 *   dm_env step types incl. the device-side auto-reset rule of the vectorised env
 *     (envs flagged in `needs_reset` on entry are FIRST: task state reset as in
 *      initialize_episode :167-174, reward 0, discount 1; on exit the flag holds the envs
 *      that terminated this step).
 * `rw` supplies the engine views, constants and the reward outputs; its task-state
 * pointers (goal_current, key_norm_state, key_activation, sustain_activation,
 * finger_current) must alias the arrays named here. */
typedef struct rp_task_advance_args {
  rp_task_reward_args rw;
  int n_lookahead;                   /* L: goal_state is [E][L+1][89] */
  int n_songs, bank_len;             /* goal_bank [n_songs][bank_len][89], finger_bank [..][88] */
  int wrong_press_termination;
  double key_threshold, sustain_threshold;   /* piano.py:31-32 */
  const int* warn;                   /* [E] engine warn flags (bit 0: bad state) */
  const void* key_qrange;            /* [88][2] */
  const void* goal_bank; const long long* finger_bank;
  const long long* song_len;         /* [n_songs] */
  long long* song_id;                /* [E] (written only in prefetch mode, see next_ready) */
  /* piano state */
  void* key_state;                   /* [E][88] clipped joint position */
  const void* sustain_state;         /* [E]     latched by before_step */
  /* task state (in/out) */
  long long* t_idx; unsigned char* should_terminate; unsigned char* failure_termination;
  void* discount_state;              /* [E] */
  void* goal_state;                  /* [E][L+1][89] */
  long long* finger_next;            /* [E][88] */
  void* fingering_state;             /* [E][10] ([E][5] with rw.hand_filter) */
  unsigned char* needs_reset;        /* [E] in/out */
  /* outputs */
  void* discount;                    /* [E] */
  int* step_type;                    /* [E] 0 FIRST, 1 MID, 2 LAST */
  /* optional (eval_sums == NULL: off): MidiEvaluationWrapper (wrappers/evaluation.py:67-177) as a
   * device reduction.  Per simulated step: precision / recall / F1 (sklearn "binary",
   * zero_division=1) of the key activations against the goal row of that step, and of the sustain
   * activation against the goal sustain; accumulated per env, and at LAST the episode mean goes into
   * a ring of the last `eval_deque` episodes.  Always double. */
  double* eval_sums;                 /* [E][6] running sums: key P, R, F1, sustain P, R, F1 */
  double* eval_count;                /* [E]    steps accumulated */
  double* eval_hist;                 /* [E][eval_deque][6] episode means */
  long long* eval_nfinished;         /* [E]    episodes finished */
  int eval_deque;
  /* optional (next_ready == NULL: off): double-buffered goal bank for per-episode MIDI
   * augmentations without a host round trip.  Env e owns bank slots 2e and 2e+1; the host keeps
   * the slot the env is not playing filled with the tables of its next episode and sets
   * next_ready[e].  When env e starts an episode and next_ready[e] is set, the launch switches
   * song_id[e] to the other slot, clears next_ready[e] and raises consumed[e] (the host polls that
   * asynchronously, refills the freed slot, clears it).  If the host is late the env replays its
   * current tables. */
  unsigned char* next_ready;         /* [E] in/out */
  unsigned char* consumed;           /* [E] out (sticky) */
  /* engine warn bits that end the episode with reward 0 / discount 0 (0 = RP_WARN_BADSTATE only);
   * fatal_count (optional) counts those episodes per env */
  int warn_fatal_mask;
  long long* fatal_count;            /* [E] or NULL */
  /* warn_count (optional) counts per env the episodes that END -- for whatever reason -- with one of
   * warn_count_mask's bits raised: the episodes a capacity overflow touched when overflows do not end them
   * (the reference ends an episode only at the end of the MIDI or on a wrong press,
   * suite/tasks/piano_with_shadow_hands.py:212-220) */
  int warn_count_mask;
  long long* warn_count;             /* [E] or NULL */
  /* optional (traj_record == NULL: off): the compact per-env trajectory record of the multi-GPU gather (SURVEY 8e,
   * robopianist_amd/distributed.py: pack_trajectory_record), written by this launch straight into the buffer the
   * all-gather sends -- no allocation and no extra launch per step on the N > 1 path.  One row per env, in the engine's
   * precision T:  qpos[nv] | reward | discount | step_type | the 88 activation bits as raw bytes (little-endian, key k
   * = bit k % 8 of byte k / 8) in the last 16 / sizeof(T) + (T == float) words, i.e. 2 (double) or 3 (float).
   * Row length = nv + 3 + (T == double ? 2 : 3). */
  void* traj_record;
} rp_task_advance_args;

int rp_task_advance(const rp_task_advance_args* args, void* hip_stream);

/* rp_task_prestep: everything between env.step(action) and physics.step(), one launch (thread = env):
 *   - CanonicalSpecWrapper.step (robopianist/wrappers/canonical.py; dm_env_wrappers): an action in [-1, 1] is mapped
 *     onto the spec's bounds, lo + (a + 1) / 2 * (hi - lo), optionally clipped to [-1, 1] first (act_lo == NULL: the
 *     action already is in the spec's units);
 *   - composer.Environment.step's bookkeeping: an env whose episode ended at the previous step (needs_reset) is
 *     reset and NOT simulated in this step, its action is discarded (dm_env): active = !needs_reset goes to the
 *     engine's RP_ACTIVE mask, reset_mask = needs_reset is what rp_step_masked takes;
 *   - PianoWithShadowHands.before_step (suite/tasks/piano_with_shadow_hands.py:176-186): the hand actions go to
 *     their actuators' ctrl (right hand's, then the left's: hand_act), the last action entry to
 *     piano.apply_sustain (models/piano/piano.py:140-143; 0 for an env that is being reset). */
typedef struct rp_task_prestep_args {
  int n_envs, precision;             /* 32 / 64: element type of action, bounds, ctrl, sustain_state */
  int n_action, nu;                  /* action row length (hand actions + sustain), ctrl row length */
  const void* action;                /* [E][n_action] */
  const void* act_lo;                /* [n_action] or NULL */
  const void* act_range;             /* [n_action] hi - lo (with act_lo) */
  int clip;
  const unsigned char* needs_reset;  /* [E] */
  const int* hand_act;               /* [n_action - 1] */
  void* ctrl;                        /* [E][nu], rows of the envs that are stepped */
  void* sustain_state;               /* [E] */
  int* active;                       /* [E] out */
  unsigned char* reset_mask;         /* [E] out */
  /* Scripted replay (optional; round 6).  The reference's example replays a recorded action table, one row per control
   * step of every episode (/root/reference/examples/piano_with_shadow_hands_env.py:110-141: `for t: env.step(actions[t])`).
   * With `action_table` set, env e takes row action_index[e] of the table instead of `action` (which may then be NULL),
   * and the launch advances the index itself: 0 for an env that is being reset (its step returns FIRST and consumes no
   * row), min(index + 1, action_table_len - 1) otherwise -- no host-side gather or index arithmetic between two steps. */
  const void* action_table;          /* [action_table_len][n_action] or NULL */
  long long* action_index;           /* [E] in / out (with action_table) */
  int action_table_len;
} rp_task_prestep_args;

int rp_task_prestep(const rp_task_prestep_args* args, void* hip_stream);

/* rp_task_rasterize: goal / fingering tables of augmented songs, built on the device.
 * What the reference does on the host at every episode start when `augmentations` are given
 * (suite/tasks/piano_with_shadow_hands.py:151-165): MidiFile.stretch / transpose
 * (music/midi_file.py:204-229), sequence_to_pianoroll (music/piano_roll.py:59-204, onset window 0)
 * and NoteTrajectory.seq_to_trajectory incl. the repeated-note gap and the sustain latch
 * (music/midi_file.py:315-362), then the tables the vectorised task reads.  One workgroup per job
 * (= one bank slot); a job names a base song and an ordered list of operations. */
typedef struct rp_task_raster_args {
  int n_jobs, precision;
  int n_songs, bank_len, max_ops, n_buffer;   /* n_buffer: rows of silence in front (initial_buffer_time) */
  double fps;                                 /* 1 / control_timestep */
  /* base songs: notes stably sorted by start time, CC64 (sustain) events in file order */
  const long long* note_ofs;                  /* [n_songs + 1] */
  const double* note_start; const double* note_end;
  const int* note_pitch; const int* note_velocity; const int* note_part;
  const long long* cc_ofs;                    /* [n_songs + 1] */
  const double* cc_time; const int* cc_value;
  const double* total_time;                   /* [n_songs] */
  /* jobs */
  const long long* job_slot;                  /* [n_jobs] bank slot to fill */
  const int* job_song;                        /* [n_jobs] base song */
  const int* op_kind;                         /* [n_jobs][max_ops] 0 = none, 1 = stretch, 2 = transpose */
  const double* op_value;                     /* [n_jobs][max_ops] factor / semitones */
  /* outputs */
  void* goal_bank;                            /* [n_slots][bank_len][89] */
  long long* finger_bank;                     /* [n_slots][bank_len][88] */
  long long* song_len;                        /* [n_slots] */
  int* status;                                /* [n_jobs] 0 = done, 1 = does not fit in bank_len rows (slot untouched) */
} rp_task_raster_args;

int rp_task_rasterize(const rp_task_raster_args* args, void* hip_stream);
const char* rp_task_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RP_TASK_H_ */

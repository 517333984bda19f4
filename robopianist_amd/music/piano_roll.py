"""Piano roll with fingering (restates robopianist/music/piano_roll.py:59-204 for the
only mode the task uses: onset_mode="window", min_frame_occupancy_for_label=0,
onset_overlap=True, no onset delay)."""

from __future__ import annotations

import collections
import math

import numpy as np

ONSET_WINDOW = 1

Pianoroll = collections.namedtuple(
    "Pianoroll",
    ["active", "onsets", "onset_velocities", "active_velocities", "control_changes",
     "fingerings"],
)


def sequence_to_pianoroll(sequence, frames_per_second, min_pitch, max_pitch,
                          max_velocity=127, onset_window=ONSET_WINDOW) -> Pianoroll:
    n_frames = int(sequence.total_time * frames_per_second + 1)  # piano_roll.py:78-81
    roll = np.zeros((n_frames, max_pitch - min_pitch + 1), dtype=np.float32)
    onsets = np.zeros_like(roll)
    control_changes = np.zeros((n_frames, 128), dtype=np.int32)
    fingerings = np.full_like(roll, -1)
    velocities_roll = np.zeros_like(roll)

    def frames_from_times(start_time, end_time):
        start_frame = int(start_time * frames_per_second)
        end_frame = int(math.ceil(end_time * frames_per_second))
        end_frame = max(start_frame + 1, end_frame)  # every note fills >= 1 frame
        return start_frame, end_frame

    for note in sorted(sequence.notes, key=lambda n: n.start_time):
        if note.pitch < min_pitch or note.pitch > max_pitch:
            continue
        start_frame, end_frame = frames_from_times(note.start_time, note.end_time)
        onset_start_wo, _ = frames_from_times(note.start_time, note.end_time)
        onset_start_frame = max(0, onset_start_wo - onset_window)
        onset_end_frame = min(onsets.shape[0], onset_start_wo + onset_window + 1)
        p = note.pitch - min_pitch
        onsets[onset_start_frame:onset_end_frame, p] = 1.0
        roll[start_frame:end_frame, p] = 1.0
        if note.velocity > max_velocity:
            raise ValueError("Note velocity exceeds max velocity: %d > %d"
                             % (note.velocity, max_velocity))
        velocities_roll[start_frame:end_frame, p] = note.velocity / max_velocity
        if note.part is not None:
            fingerings[start_frame:end_frame, p] = note.part

    for cc in sequence.control_changes:
        frame, _ = frames_from_times(cc.time, 0)
        if frame < len(control_changes):
            control_changes[frame, cc.control_number] = cc.control_value + 1

    return Pianoroll(active=roll, onsets=onsets, onset_velocities=velocities_roll * onsets,
                     active_velocities=velocities_roll, control_changes=control_changes,
                     fingerings=fingerings)

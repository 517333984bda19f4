"""Music constants (mirror of robopianist/music/constants.py:17-59)."""

MIN_MIDI_PITCH = 0
MAX_MIDI_PITCH = 127
MIN_MIDI_PITCH_PIANO = 21  # A0
MAX_MIDI_PITCH_PIANO = 108  # C8
MIN_KEY_NUMBER = 0
MAX_KEY_NUMBER = 87
NUM_KEYS = MAX_KEY_NUMBER - MIN_KEY_NUMBER + 1

NOTES_IN_OCTAVE = ["C", "C#", "D", "D#", "E", "F", "F#", "G", "G#", "A", "A#", "B"]
NOTES = ["A0", "A#0", "B0"]
for _octave in range(1, 8):
    for _note in NOTES_IN_OCTAVE:
        NOTES.append(_note + str(_octave))
NOTES.append("C8")
assert len(NOTES) == NUM_KEYS

KEY_NUMBER_TO_NOTE_NAME = {i: note for i, note in enumerate(NOTES)}
NOTE_NAME_TO_KEY_NUMBER = {note: i for i, note in enumerate(NOTES)}
MIDI_NUMBER_TO_NOTE_NAME = {i + 21: name for i, name in enumerate(NOTES)}
NOTE_NAME_TO_MIDI_NUMBER = {v: k for k, v in MIDI_NUMBER_TO_NOTE_NAME.items()}

SAMPLING_RATE = 44100
SUSTAIN_PEDAL_CC_NUMBER = 64
MIN_CC_VALUE = 0
MAX_CC_VALUE = 127
MIN_VELOCITY = 0
MAX_VELOCITY = 127

"""Music module (mirror of robopianist/music/__init__.py:57-92)."""

from pathlib import Path
from typing import Union

from robopianist_amd.music import library, midi_file

# The PIG dataset is licence-gated and absent (docs/dataset.md); only the debug
# songs ship, exactly like a fresh checkout of the reference.
PIG_MIDIS: list = []
ETUDE_MIDIS: list = []
DEBUG_MIDIS = list(library.MIDI_NAME_TO_CALLABLE.keys())
ALL = DEBUG_MIDIS + PIG_MIDIS


def load(path_or_name: Union[str, Path], stretch: float = 1.0, shift: int = 0) -> midi_file.MidiFile:
    """Make a MidiFile from a path or a library name.

    Raises ValueError for unsupported extensions / invalid files and KeyError for
    unknown names, like the reference.
    """
    path = Path(path_or_name)
    if path.suffix:
        midi = midi_file.MidiFile.from_file(path)
    else:
        if path.stem in DEBUG_MIDIS:
            midi = library.MIDI_NAME_TO_CALLABLE[path.stem]()
        else:
            raise KeyError(f"Unknown name: {path.stem}. Available names: {ALL}.")
    return midi.stretch(stretch).transpose(shift)


__all__ = ["ALL", "DEBUG_MIDIS", "PIG_MIDIS", "ETUDE_MIDIS", "load"]

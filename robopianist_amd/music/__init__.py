"""Music module (mirror of robopianist/music/__init__.py:57-92)."""

import os
from pathlib import Path
from typing import Union

from robopianist_amd.music import library, midi_file


def _camel_case(name: str) -> str:
    """music/__init__.py:22-30: 'nocturne_op_9-1' -> 'NocturneOp9' (the -N suffix distinguishes the
    fingering annotations of one piece in the PIG dataset)."""
    new_name = name.replace("'", "")
    new_name = new_name.replace("_", " ").title().replace(" ", "")
    if "-" in new_name:
        new_name = new_name[: new_name.index("-")]
    return new_name


# The PIG dataset is licence-gated (docs/dataset.md) and does not ship; a user who has it points
# ROBOPIANIST_PIG_DIR at the directory of `*.proto` NoteSequences (or drops it at
# music/data/pig_single_finger, the reference's location) and the repertoire names appear.
_PIG_DIR = Path(os.environ.get("ROBOPIANIST_PIG_DIR", Path(__file__).parent / "data" / "pig_single_finger"))
_PIG_FILES = sorted(_PIG_DIR.glob("*.proto")) if _PIG_DIR.is_dir() else []
PIG_MIDIS = [_camel_case(Path(f).stem) for f in _PIG_FILES]
_PIG_NAME_TO_FILE = dict(zip(PIG_MIDIS, _PIG_FILES))
_ETUDE_SUBSET = (  # music/__init__.py:36-49
    "french_suite_no_1_allemande-1", "french_suite_no_5_sarabande-1", "piano_sonata_d_845_1st_mov-1",
    "partita_no_2_6-1", "waltz_op_64_no_1-1", "bagatelle_op_3_no_4-1", "kreisleriana_op_16_no_8-1",
    "french_suite_no_5_gavotte-1", "piano_sonata_no_23_2nd_mov-1", "golliwogg's_cakewalk-1",
    "piano_sonata_no_2_1st_mov-1", "piano_sonata_k_279_in_c_major_1st_mov-1",
)
# (the reference lists the etudes unconditionally; loading one without the dataset raises KeyError)
ETUDE_MIDIS = [_camel_case(name) for name in _ETUDE_SUBSET] if PIG_MIDIS else []
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
        elif path.stem in _PIG_NAME_TO_FILE:
            midi = midi_file.MidiFile.from_file(_PIG_NAME_TO_FILE[path.stem])
        else:
            raise KeyError(f"Unknown name: {path.stem}. Available names: {ALL}.")
    return midi.stretch(stretch).transpose(shift)


__all__ = ["ALL", "DEBUG_MIDIS", "PIG_MIDIS", "ETUDE_MIDIS", "load"]

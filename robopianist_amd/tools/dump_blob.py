"""Writes the model blob `rp_create` consumes (model constants + engine tables) to a file, so that
a non-Python host can drive the C ABI:  python -m robopianist_amd.tools.dump_blob scene.blob"""
import argparse
import warnings

from robopianist_amd import engine
from robopianist_amd.model import scene


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--hands", default="right,left", help="comma list of: right, left (empty: piano only)")
    ap.add_argument("--gravity_compensation", action="store_true")
    ap.add_argument("--reduced_action_space", action="store_true")
    args = ap.parse_args()
    hands = tuple(h for h in args.hands.split(",") if h)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        si = scene.build_scene(hands=hands, gravity_compensation=args.gravity_compensation,
                               reduced_action_space=args.reduced_action_space,
                               primitive_fingertip_collisions=True)
    blob = engine.make_blob(si.model, si.key_joint_ids)
    with open(args.out, "wb") as f:
        f.write(blob)
    print(f"{args.out}: {len(blob)} bytes, nv={si.model.nv} nu={si.model.nu}")


if __name__ == "__main__":
    main()

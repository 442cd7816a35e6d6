"""Drop-in for the reference's run_valley_llamma_v2.py (a script that runs at import there; a function here).
Usage: python -m valley.inference.run_valley_llamma_v2 <video_file>"""
import argparse

from valley_amd.cli import SAMPLED as gen_kwargs, SYSTEM_TURN as system_prompt, VALLEY2_7B, init_vision_token, main_v2, v2_message  # noqa: F401

if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Process some video.")
    parser.add_argument("video_file", type=str, help="The path to the video file")
    print(main_v2(parser.parse_args().video_file))

"""Drop-in for the reference's run_valley_conv.py: implementation in valley_amd/cli.py."""
from valley_amd.cli import assistant_out, conv_inference as inference, conv_parse_args  # noqa: F401
from valley_amd.video import load_video  # noqa: F401

if __name__ == "__main__":
    inference(conv_parse_args())

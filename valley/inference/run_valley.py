"""Drop-in for the reference's entry point: the implementation lives in valley_amd/cli.py."""
from valley_amd.cli import SYSTEM_TURN as DEFAULT_SYSTEM, init_vision_token, main, parse_args  # noqa: F401

if __name__ == "__main__":
    main(parse_args())

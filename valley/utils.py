"""reference valley/utils.py:146-152 — the only helper the inference entry points use."""


def disable_torch_init():
    """The reference skips torch's default nn.Linear/LayerNorm init to speed model creation; the HIP
    engines never run torch initialisers, so this is a no-op kept for call-site compatibility."""
    return None

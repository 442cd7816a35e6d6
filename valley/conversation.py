"""Same import path as the reference's valley/conversation.py; implementation in valley_amd."""
from valley_amd.conversation import Conversation, SeparatorStyle, conv_templates, default_conversation  # noqa: F401

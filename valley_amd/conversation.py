"""Prompt assembly for the interactive entry point (reference valley/conversation.py:13-48 ``get_prompt`` and the two
templates :202-228).  Only what ``run_valley_conv.py`` touches: roles, separator styles, message list, ``copy()``.  The
gradio / base64 media helpers of the reference class belong to its web UI (out of scope)."""
from __future__ import annotations

from dataclasses import dataclass, field
from enum import Enum, auto
from typing import List, Optional, Sequence


class SeparatorStyle(Enum):
    SINGLE = auto()
    TWO = auto()


@dataclass
class Conversation:
    system: str
    roles: Sequence[str]
    messages: List[List[str]] = field(default_factory=list)
    offset: int = 0
    sep_style: SeparatorStyle = SeparatorStyle.SINGLE
    sep: str = "###"
    sep2: Optional[str] = None
    has_video: bool = False                      # set by the chat loop once the visual block has been sent

    def get_prompt(self) -> str:
        """system + sep, then ``role: message + sep`` per turn (an empty message leaves ``role:`` open); with
        SeparatorStyle.TWO the separators alternate sep / sep2."""
        if self.sep_style not in (SeparatorStyle.SINGLE, SeparatorStyle.TWO):
            raise ValueError(f"Invalid style: {self.sep_style}")
        seps = (self.sep, self.sep) if self.sep_style == SeparatorStyle.SINGLE else (self.sep, self.sep2)
        out = [self.system, seps[0]]
        for i, (role, message) in enumerate(self.messages):
            if isinstance(message, tuple):        # (text, media path, mode) triples of the web UI
                message = message[0]
            out.append(f"{role}: {message}{seps[i % 2]}" if message else f"{role}:")
        return "".join(out)

    def append_message(self, role: str, message) -> None:
        self.messages.append([role, message])

    def copy(self) -> "Conversation":
        return Conversation(self.system, self.roles, [[r, m] for r, m in self.messages], self.offset, self.sep_style,
                            self.sep, self.sep2)


conv_templates = {
    "v1": Conversation(
        system="A chat between a curious human and an artificial intelligence assistant. "
               "The assistant gives helpful, detailed, and polite answers to the human's questions.",
        roles=("Human", "Assistant")),
    "multimodal_video": Conversation(
        # the reference concatenates these three sentences WITHOUT spaces (conversation.py:213-215)
        system="You are Valley, a large language and vision assistant trained by ByteDance."
               "You are able to understand the visual content or video that the user provides, and assist the user with a "
               "variety of tasks using natural language."
               "Follow the instructions carefully and explain your answers in detail.",
        roles=("Human", "Assistant"),
        messages=[["Human", "Hi!"], ["Assistant", "Hi there!  How can I help you today?\n"]], offset=2),
}
default_conversation = conv_templates["multimodal_video"]

"""Prompt templates.

Produces prompt strings byte-identical to the reference's conversation
templates (reference: dataset/conversation.py:19-237) so that token ids match
for a fixed tokenizer.  The design differs: a template is an immutable
``PromptTemplate`` plus a per-style *renderer function* looked up in a table,
and a ``Conversation`` is just (template, list of turns).
"""
from __future__ import annotations

import base64
import dataclasses
from enum import Enum, auto
from io import BytesIO
from typing import Callable, Dict, List, Optional, Sequence, Tuple

from ..constants import (DEFAULT_EV_END_TOKEN, DEFAULT_EV_START_TOKEN,
                         DEFAULT_EVENT_TOKEN)


class SeparatorStyle(Enum):
    SINGLE = auto()
    TWO = auto()
    MPT = auto()
    PLAIN = auto()
    LLAMA_2 = auto()


def _text_of(message):
    """A turn's payload is either a string or ``(text, image, mode)``."""
    if isinstance(message, tuple):
        return message[0]
    return message


def _render_single(conv, turns):
    out = conv.system + conv.sep
    for role, msg in turns:
        out += f"{role}: {_text_of(msg)}{conv.sep}" if msg else f"{role}:"
    return out


def _render_two(conv, turns):
    seps = (conv.sep, conv.sep2)
    out = conv.system + seps[0]
    for i, (role, msg) in enumerate(turns):
        out += f"{role}: {_text_of(msg)}{seps[i % 2]}" if msg else f"{role}:"
    return out


def _render_mpt(conv, turns):
    out = conv.system + conv.sep
    for role, msg in turns:
        out += f"{role}{_text_of(msg)}{conv.sep}" if msg else role
    return out


def _render_llama2(conv, turns):
    out = ""
    for i, (role, msg) in enumerate(turns):
        if i == 0:
            if not msg:
                raise ValueError("the first turn of a LLAMA_2 prompt must carry text")
            if role != conv.roles[0]:
                raise ValueError("the first turn of a LLAMA_2 prompt must be the user's")
        if not msg:
            continue
        text = _text_of(msg)
        if i == 0 and conv.system:
            text = f"<<SYS>>\n{conv.system}\n<</SYS>>\n\n{text}"
        if i % 2 == 0:
            out += f"{conv.sep}[INST] {text} [/INST]"
        else:
            out += f" {text} {conv.sep2}"
    return out.lstrip(conv.sep)


def _render_plain(conv, turns):
    seps = (conv.sep, conv.sep2)
    out = conv.system
    for i, (_, msg) in enumerate(turns):
        if msg:
            out += _text_of(msg) + seps[i % 2]
    return out


_RENDERERS: Dict[SeparatorStyle, Callable] = {
    SeparatorStyle.SINGLE: _render_single,
    SeparatorStyle.TWO: _render_two,
    SeparatorStyle.MPT: _render_mpt,
    SeparatorStyle.LLAMA_2: _render_llama2,
    SeparatorStyle.PLAIN: _render_plain,
}


@dataclasses.dataclass
class Conversation:
    """Conversation history + the template that renders it."""

    system: str
    roles: Sequence[str]
    messages: List[List]
    offset: int = 0
    sep_style: SeparatorStyle = SeparatorStyle.SINGLE
    sep: str = "###"
    sep2: Optional[str] = None
    version: str = "Unknown"
    skip_next: bool = False

    # ------------------------------------------------------------------ prompt
    def _normalised_turns(self) -> List[Tuple[str, object]]:
        """First-turn image tuples get the event tag moved to the front."""
        turns = [tuple(m) for m in self.messages]
        if turns and isinstance(turns[0][1], tuple):
            role, payload = turns[0]
            text = payload[0].replace(DEFAULT_EVENT_TOKEN, "").strip()
            if "mmtag" in self.version:
                turns[0] = (role, text)
                turns.insert(0, (self.roles[0], "<Event><event></Event>"))
                turns.insert(1, (self.roles[1], "Received."))
            else:
                turns[0] = (role, DEFAULT_EVENT_TOKEN + "\n" + text)
        return turns

    def get_prompt(self) -> str:
        try:
            render = _RENDERERS[self.sep_style]
        except KeyError:
            raise ValueError(f"Invalid style: {self.sep_style}")
        return render(self, self._normalised_turns())

    def append_message(self, role, message) -> None:
        self.messages.append([role, message])

    def copy(self) -> "Conversation":
        return Conversation(
            system=self.system, roles=self.roles,
            messages=[[r, m] for r, m in self.messages], offset=self.offset,
            sep_style=self.sep_style, sep=self.sep, sep2=self.sep2,
            version=self.version)

    # ------------------------------------------------------------- image utils
    @staticmethod
    def process_image(image, image_process_mode, return_pil=False,
                      image_format="PNG", max_len=1344, min_len=672):
        """Pad / resize a PIL image for display or transport.

        Mirrors the modes accepted by the reference's demo helper
        (reference: dataset/conversation.py:113-151).
        """
        from PIL import Image
        if image_process_mode == "Pad":
            w, h = image.size
            if w != h:
                side = max(w, h)
                canvas = Image.new(image.mode, (side, side), (122, 116, 104))
                canvas.paste(image, ((side - w) // 2, (side - h) // 2))
                image = canvas
        elif image_process_mode == "Resize":
            image = image.resize((336, 336))
        elif image_process_mode not in ("Default", "Crop"):
            raise ValueError(f"Invalid image_process_mode: {image_process_mode}")
        if max(image.size) > max_len:
            long_e, short_e = max(image.size), min(image.size)
            ratio = long_e / short_e
            new_short = int(min(max_len / ratio, min_len, short_e))
            new_long = int(new_short * ratio)
            w, h = image.size
            image = image.resize((new_short, new_long) if h > w else (new_long, new_short))
        if return_pil:
            return image
        buf = BytesIO()
        image.save(buf, format=image_format)
        return base64.b64encode(buf.getvalue()).decode()

    def get_images(self, return_pil=False):
        out = []
        for i, (_, msg) in enumerate(self.messages[self.offset:]):
            if i % 2 == 0 and isinstance(msg, tuple):
                _, image, mode = msg
                out.append(self.process_image(image, mode, return_pil=return_pil))
        return out

    def to_gradio_chatbot(self):
        rows = []
        for i, (_, msg) in enumerate(self.messages[self.offset:]):
            if i % 2:
                rows[-1][-1] = msg
                continue
            if isinstance(msg, tuple):
                text, image, _ = msg
                b64 = self.process_image(image, "Default", image_format="JPEG")
                tag = f'<img src="data:image/jpeg;base64,{b64}" alt="user upload image" />'
                msg = tag + text.replace("<image>", "").strip()
            rows.append([msg, None])
        return rows

    def dict(self):
        msgs = self.messages
        if self.get_images():
            msgs = [[r, m[0] if isinstance(m, tuple) else m] for r, m in msgs]
        return {"system": self.system, "roles": self.roles, "messages": msgs,
                "offset": self.offset, "sep": self.sep, "sep2": self.sep2}


conv_eventgpt_v1 = Conversation(
    system=("A chat between a curious human and an artificial intelligence assistant. "
            "The assistant gives helpful, detailed, and polite answers to the human's questions."),
    roles=("USER", "ASSISTANT"),
    version="v1",
    messages=[],
    offset=0,
    sep_style=SeparatorStyle.TWO,
    sep=" ",
    sep2="</s>",
)

default_conversation = conv_eventgpt_v1
conv_templates = {"eventgpt_v1": conv_eventgpt_v1}


def prepare_event_prompt(query: str, conv_mode: str = "eventgpt_v1") -> str:
    """``<ev_start><event><ev_end>\\n{query}`` as the user turn, empty assistant turn."""
    conv = conv_templates[conv_mode].copy()
    conv.append_message(conv.roles[0],
                        DEFAULT_EV_START_TOKEN + DEFAULT_EVENT_TOKEN + DEFAULT_EV_END_TOKEN
                        + "\n" + query)
    conv.append_message(conv.roles[1], None)
    return conv.get_prompt()

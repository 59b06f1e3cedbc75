"""eventgpt_amd — an MI355X-native event-stream multimodal LLM framework."""
__version__ = "0.1.0"

from .constants import *  # noqa: F401,F403

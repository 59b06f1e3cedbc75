from .config import ClipVisionConfig, EventChatConfig
from .clip_vit import ClipVisionModel
from .llama import LlamaModel, kv_cache_bytes
from .event_chat import (EventChatLlamaModel, EventChatModel, GenerateOutput, MLPProjector,
                         VisualTower, get_spatio_temporal_features)

__all__ = ["ClipVisionConfig", "EventChatConfig", "ClipVisionModel", "LlamaModel",
           "EventChatLlamaModel", "EventChatModel", "GenerateOutput", "MLPProjector",
           "VisualTower", "get_spatio_temporal_features", "kv_cache_bytes"]

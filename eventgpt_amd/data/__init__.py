from .conversation import (Conversation, SeparatorStyle, conv_templates, default_conversation,
                           prepare_event_prompt)
from .events import (event_frames, load_event_npy, load_image, rasterize, save_event_npy,
                     split_by_count, split_by_time, synthetic_events, voxel_grid)
from .tokenizer import ByteTokenizer, load_tokenizer, tokenizer_event_token
from .clip_preprocess import EventImageProcessor, preprocess_image

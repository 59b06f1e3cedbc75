"""Framework-wide constants.

The sentinel values and special-token strings are part of the on-disk /
prompt-level contract shared with checkpoints trained by the reference
(reference: dataset/constants.py:7-13), so they are reproduced by value.
"""

# Label value ignored by the language-model loss.
IGNORE_INDEX = -100
# Placeholder id spliced into ``input_ids`` where the event features go.
EVENT_TOKEN_INDEX = -200

DEFAULT_EVENT_TOKEN = "<event>"
DEFAULT_EVENT_PATCH_TOKEN = "<ev_patch>"
DEFAULT_EV_START_TOKEN = "<ev_start>"
DEFAULT_EV_END_TOKEN = "<ev_end>"
EVENT_PLACEHOLDER = "<event-placeholder>"

# Hard cap on the multimodal sequence length (reference:
# model/EventChatModel.py:415-418).
MAX_MULTIMODAL_LENGTH = 2048

# Geometry of the event vision tower (CLIP ViT-L/14 @ 336 px).
CLIP_IMAGE_SIZE = 336
CLIP_PATCH_SIZE = 14
CLIP_NUM_PATCHES = (CLIP_IMAGE_SIZE // CLIP_PATCH_SIZE) ** 2  # 576
CLIP_NUM_TOKENS = CLIP_NUM_PATCHES + 1                         # 577 (CLS + patches)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

# Number of equal-count temporal slices an event stream is cut into.
DEFAULT_NUM_EVENT_FRAMES = 5

from .image_utils import add_break_line, calculate_vq_params  # noqa: F401
from .prompt_utils import generate_text_image_to_text_image_prompt  # noqa: F401
from .sequence import SPECIAL_TOKENS, build_ti2ti_sequence  # noqa: F401

from .image_utils import (add_break_line, calculate_vq_params, decode_step_preview, decode_vq_to_image,  # noqa: F401
                          encode_img_with_breaks, encode_img_with_paint)
from .prompt_utils import generate_text_image_to_text_image_prompt  # noqa: F401
from .sequence import SPECIAL_TOKENS, build_ti2ti_sequence  # noqa: F401

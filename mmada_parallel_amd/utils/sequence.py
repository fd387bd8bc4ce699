"""Interleaved TI2TI token sequence, assembled exactly like the reference CLI (inference.py:117-161).

    con   = prompt[:-1] + [BOI img.. EOI] + prompt[-1:] + [BOA, BOI] + (MASK x W + NL) x H + [EOI] + MASK x T + </answer>
    uncon_text  = uncon_prompt[:-1] + img + uncon_prompt[-1:]          uncon_image = prompt_ids (text only)
    image_start = len(con_prefix) + 2 ; text_start = image_end + 1 ; text_end = text_start + T
"""
from typing import Dict, List, Optional

from .image_utils import add_break_line, calculate_vq_params

SPECIAL_TOKENS = {  # inference.py:22-31
    "mask_token": 126336, "newline_token": 126084, "image_token_offset": 126356, "answer_start": 126354,
    "answer_end": 126355, "boi": 126349, "eoi": 126350, "uncondition": 126351,
}


def build_ti2ti_sequence(prompt_ids: List[int], uncon_text_ids: List[int], input_img_token: List[int], height: int,
                         width: int, text_gen_length: int, end_token_ids: List[int], vae_scale: int = 16,
                         img_mask_token: Optional[List[int]] = None) -> Dict:
    """Returns the ids and offsets `generate_ti2ti` is called with (inference.py:129-161, 169-193).

    `input_img_token` is the encoded conditioning image incl. <boi>/<eoi>/newlines (encode_img_with_breaks);
    `img_mask_token` overrides the all-MASK output grid (painting mode, :141-146)."""
    MASK, NL = SPECIAL_TOKENS["mask_token"], SPECIAL_TOKENS["newline_token"]
    con_input_list = prompt_ids[:-1] + input_img_token + prompt_ids[-1:]
    uncon_input_text = uncon_text_ids[:-1] + input_img_token + uncon_text_ids[-1:]
    uncon_input_image = list(prompt_ids)
    seq_len, newline_every, gh, gw = calculate_vq_params(height, width, vae_scale)
    if img_mask_token is None:
        img_mask_token = add_break_line([MASK] * seq_len, gh, gw, new_number=NL)
    pred_token = ([SPECIAL_TOKENS["answer_start"], SPECIAL_TOKENS["boi"]] + img_mask_token + [SPECIAL_TOKENS["eoi"]]
                  + [MASK] * text_gen_length + list(end_token_ids))
    image_start = len(con_input_list) + 2
    image_end = image_start + len(img_mask_token)
    text_start = image_end + 1
    return dict(input_ids=con_input_list + pred_token, uncon_text=uncon_input_text, uncon_image=uncon_input_image,
                code_start=len(con_input_list), image_start=image_start, image_end=image_end, text_start=text_start,
                text_end=text_start + text_gen_length, seq_len=seq_len, newline_every=newline_every)

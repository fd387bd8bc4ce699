"""Pure list helpers of the reference's utils/image_utils.py that the TI2TI sequence builder needs
(`calculate_vq_params` :95-111, `add_break_line` :149-157).  The VQ-VAE encode/decode functions of that file wrap
third-party `diffusers` models and are out of scope."""
from typing import List


def calculate_vq_params(image_height: int, image_width: int, vae_scale: int = 16):
    """-> (seq_len, newline_every, token_grid_height, token_grid_width)."""
    token_grid_height = image_height // vae_scale
    token_grid_width = image_width // vae_scale
    return token_grid_height * token_grid_width, token_grid_width, token_grid_height, token_grid_width


def add_break_line(sequence: List[int], H: int, W: int, new_number: int = 0) -> List[int]:
    """Append `new_number` after every row of W tokens of an H x W grid."""
    result: List[int] = []
    for i in range(H):
        result.extend(sequence[i * W:(i + 1) * W] + [new_number])
    return result

"""Prompt template of the TI2TI task (reference utils/prompt_utils.py:209-233): strings only."""


def generate_text_image_to_text_image_prompt(prompt_text: str, system_prompt: str):
    """-> (conditional prompt, unconditional prompt) — the latter replaces the instruction by `<uncondition>`."""
    head = f"<system>{system_prompt}</system>"
    return head + f"<user>{prompt_text}</user>", head + "<user><uncondition></user>"

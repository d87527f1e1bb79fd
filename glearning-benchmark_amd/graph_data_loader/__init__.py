"""Drop-in for the reference's `graph_data_loader` package on the tokenizer path: put the parent
directory (glearning-benchmark_amd/) ahead of the reference on sys.path and
`from graph_data_loader import TokenDataset, collate, ...` resolves here.
The MPNN/GPS-only names (GraphTokenDataset, AddQueryEncoding, balance_classes, ...) are out of scope:
they never reach a tokenizer."""
from .data_loader import (SPECIAL, TokenDataset, build_vocab_from_texts, build_vocab_from_texts_on_device, collate, determine_num_classes,  # noqa: F401
                          determine_num_classes_pyg, load_examples, load_examples_multi_algorithm,
                          parse_distance_label_from_text, parse_query_nodes_from_text, parse_yes_no_from_text,
                          resolve_split_globs)
from .graph_token_dataset_autograph import (GraphTokenDatasetForAutoGraph, parse_graph_from_json,  # noqa: F401
                                            parse_graph_from_text, parse_label_from_text)
from .zinc_dataset_autograph import ZINCDatasetForAutoGraph, get_zinc_num_types  # noqa: F401
from .zinc_dataset_indexbase import ZINCTokenizationDataset, collate_zinc_batch  # noqa: F401
from .zinc_vocab import (NUM_ATOM_TYPES, NUM_BOND_TYPES, ZINC_ATOM_TYPES, ZINC_BOND_TYPES,  # noqa: F401
                         build_fixed_zinc_vocab, build_zinc_vocab_on_device, extend_vocab_with_dynamic_tokens, get_atom_type_from_id,
                         get_atom_type_id, get_bond_type_from_id, get_bond_type_id, map_autograph_token_to_fixed_id)

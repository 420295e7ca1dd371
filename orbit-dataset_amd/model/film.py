"""FiLM bookkeeping for the native extractors.

Mirror of reference model/film.py:38-94. Which BatchNorms are modulated is decided by the native plan
(csrc/extractor.hip follows film.py:41-48 for EfficientNet: root bn1/bn2 and InvertedResidual.bn2; every
BatchNorm for the build-added resnet18); this module marks them with `.film = True` exactly as the
reference does, and derives names / sizes / initial values from the module tree the same way.
"""


def tag_film_layers(feature_extractor_name, feature_extractor):
    for _, module in feature_extractor.film_slot_modules():
        module.film = True


def get_film_parameter_names(feature_extractor_name, feature_extractor):
    parameter_list = []
    for name, module in feature_extractor.named_modules():
        if hasattr(module, "film"):
            parameter_list.append(name + ".weight")
            parameter_list.append(name + ".bias")
    return parameter_list


def unfreeze_film(film_parameter_names, feature_extractor):
    for name, param in feature_extractor.named_parameters():
        if name in film_parameter_names:
            param.requires_grad = True


def get_film_parameters(film_parameter_names, feature_extractor):
    film_params = {}
    if film_parameter_names is not None:
        for name, param in feature_extractor.named_parameters():
            if name in film_parameter_names:
                film_params[name] = param.detach().clone()
    return film_params


def get_film_parameter_sizes(film_parameter_names, feature_extractor):
    sizes = {}
    for name, param in feature_extractor.named_parameters():
        if name in film_parameter_names:
            sizes[name] = len(param)
    return sizes

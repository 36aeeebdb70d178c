from .homophily_learner import HomophilyLearner

REGISTRY = {"homophily_learner": HomophilyLearner}
